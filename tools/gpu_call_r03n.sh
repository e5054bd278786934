cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r03n
mkdir -p $O
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; rc=$?; tail -2 $O/smoke.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o z -- python3 bench.py --no-cpu-baseline > $O/prof_bench.json 2> $O/prof_bench.err; echo "prof rc=$?"
head -12 $O/prof_bench/z_kernel_stats.csv | cut -c1-130
