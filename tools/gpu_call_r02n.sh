cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r02n
mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; rc=$?; tail -3 $O/smoke.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 900 python -m pytest tests/test_reference_integration.py tests/test_p256_gpu.py -m gpu -x -q -s > $O/pytest_integ.log 2>&1; rc=$?; tail -12 $O/pytest_integ.log; [ $rc -eq 0 ] || exit 1
