cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r02p
mkdir -p $O
timeout -k 10 300 python tools/bench_zk.py 1 5 --mdoc-sig > $O/zk_mdoc_sig.json 2> $O/zk_mdoc_sig.err; rc=$?; tail -3 $O/zk_mdoc_sig.err; cat $O/zk_mdoc_sig.json; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sig -o z -- python3 tools/bench_zk.py 1 3 --mdoc-sig > $O/prof_sig.json 2> $O/prof_sig.err; echo "prof rc=$?"
head -30 $O/prof_sig/z_kernel_stats.csv | cut -c1-150
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -6 $O/pytest_gpu.log; [ $rc -eq 0 ] || exit 1
