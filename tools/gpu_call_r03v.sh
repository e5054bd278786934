cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r03v
mkdir -p $O
LFGPU_VERBOSE=1 LFGPU_P256_GRID_MAX=131072 timeout -k 10 120 python tools/bench_zk.py 1 2 --mdoc-sig > $O/zk.json 2> $O/zk.err; grep "grid:" $O/zk.err | tail -21
