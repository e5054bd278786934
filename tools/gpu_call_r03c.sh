cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r03c
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > $O/pytest_gpu.log 2>&1; rc=$?; grep -E "mdoc end to end" $O/pytest_gpu.log | cut -c1-1100; tail -3 $O/pytest_gpu.log; [ $rc -eq 0 ] || exit 1
