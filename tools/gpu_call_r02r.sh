cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r02r
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_reference_integration.py -m gpu -x -q -s -k "mdoc_end" > $O/pytest_mdoc_e2e.log 2>&1; rc=$?; tail -12 $O/pytest_mdoc_e2e.log; [ $rc -eq 0 ] || exit 1
