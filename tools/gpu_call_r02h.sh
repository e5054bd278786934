cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r02h
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lch14" > $O/pytest2.log 2>&1; rc=$?; tail -3 $O/pytest2.log; [ $rc -eq 0 ] || exit 1
for w in "2 2" "4 4" "4 2" "2 4"; do
  set -- $w
  LFGPU_BS_CIN_WPC=$1 LFGPU_BS_COUT_WPC=$2 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/q$1$2 -o z -- python3 tools/bench_lch.py 1024 20 5 > $O/q$1$2.log 2>&1 || exit 1
  echo "CIN_WPC=$1 COUT_WPC=$2"; grep "rows" $O/q$1$2.log; grep "bs_c" $O/q$1$2/z_kernel_stats.csv | cut -d, -f1-4 | sed 's/(elt_t.*"/"/; s/(unsigned.*"/"/' | cut -c1-90
done
