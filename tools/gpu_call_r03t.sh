cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r03t
mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_zk_cxx.py -m gpu -x -q -k "commitment_received" > $O/pytest_gpu.log 2>&1; rc=$?; tail -4 $O/pytest_gpu.log; [ $rc -eq 0 ] || exit 1
