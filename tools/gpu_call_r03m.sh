cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r03m
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_zk_cxx.py -m gpu -x -q -k "small_p256" > $O/pytest.log 2>&1; rc=$?; tail -15 $O/pytest.log; [ $rc -eq 0 ] || exit 1
