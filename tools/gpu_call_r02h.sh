cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r02h
mkdir -p $O
for w in 2 3 4; do
  LFGPU_BS_V2_NB=4 LFGPU_BS_CIN_WPC=$w LFGPU_BS_COUT_WPC=$w timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/w$w -o z -- python3 tools/bench_lch.py 1024 20 5 > $O/w$w.log 2>&1 || exit 1
  echo "WPC=$w"; grep "rows" $O/w$w.log; grep "bs_c" $O/w$w/z_kernel_stats.csv | cut -d, -f1-4 | sed 's/(elt_t.*"/"/; s/(unsigned.*"/"/' | cut -c1-90
done
