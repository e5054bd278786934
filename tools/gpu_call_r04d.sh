cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r04d
mkdir -p $O
for i in 1 2; do
timeout -k 10 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu_$i.log 2>&1; rc=$?; tail -2 $O/pytest_gpu_$i.log; [ $rc -eq 0 ] || exit 1
done
