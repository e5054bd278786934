cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r04b
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_hash -o z -- python3 tools/bench_zk.py 0 3 --mdoc > $O/prof_hash.json 2> $O/prof_hash.err; echo "prof rc=$?"
LFGPU_VERBOSE=1 timeout -k 10 300 python tools/bench_zk.py 0 2 --mdoc > $O/zk.json 2> $O/zk.err; grep "sumcheck_layer:" $O/zk.err | tail -17 | cut -c1-220
