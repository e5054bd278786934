cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r03z
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sig -o z -- python3 tools/bench_zk.py 1 3 --mdoc-sig > $O/prof_sig.json 2> $O/prof_sig.err; echo "prof rc=$?"
head -8 $O/prof_sig/z_kernel_stats.csv | cut -c1-140
