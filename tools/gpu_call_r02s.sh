cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r02s
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_zk_cxx.py -m gpu -x -q -s -k "mdoc_sig" > $O/pytest_sig.log 2>&1; rc=$?; tail -12 $O/pytest_sig.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 900 python -m pytest tests/test_reference_integration.py -m gpu -x -q -s -k "mdoc_end" > $O/pytest_mdoc_e2e.log 2>&1; rc=$?; tail -12 $O/pytest_mdoc_e2e.log; [ $rc -eq 0 ] || exit 1
