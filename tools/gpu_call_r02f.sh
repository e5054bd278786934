cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r02f
mkdir -p $O
for pr in 0 1 2; do
  LFGPU_BS_PROBE=$pr timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/probe$pr -o z -- python3 tools/bench_lch.py 1024 20 5 > $O/probe$pr.log 2>&1 || exit 1
  echo "probe $pr"; grep "bs_" $O/probe$pr/z_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
done
