cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r03j
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rs_encode" > $O/pytest_rs.log 2>&1; rc=$?; tail -4 $O/pytest_rs.log; [ $rc -eq 0 ] || exit 1
