cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r04c
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_zk_cxx.py tests/test_reference_integration.py tests/test_cxx_example.py -m gpu -x -q -s > $O/pytest.log 2>&1; rc=$?; grep "mdoc end to end" $O/pytest.log | cut -c1-420; tail -3 $O/pytest.log; [ $rc -eq 0 ] || exit 1
