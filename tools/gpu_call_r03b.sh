cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r03b
mkdir -p $O
LFGPU_VERBOSE=1 timeout -k 10 300 python tools/bench_zk.py 0 2 --mdoc > $O/zk_mdoc.json 2> $O/zk_mdoc.err; rc=$?; grep -E "circuit_from_lfc1|quad_upload" $O/zk_mdoc.err | head -30; [ $rc -eq 0 ] || { tail -5 $O/zk_mdoc.err; exit 1; }
