cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r02o
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_zk_cxx.py -m gpu -x -q -s -k "mdoc" > $O/pytest_mdoc.log 2>&1; rc=$?; tail -25 $O/pytest_mdoc.log; [ $rc -eq 0 ] || exit 1
