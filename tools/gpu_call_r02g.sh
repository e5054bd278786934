cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r02g
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lch14 or rs_encode" > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log; [ $rc -eq 0 ] || exit 1
for nb in 5 4; do
  LFGPU_BS_V2_NB=$nb timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/nb$nb -o z -- python3 tools/bench_lch.py 1024 20 5 > $O/nb$nb.log 2>&1 || exit 1
  echo "V2 NB=$nb"; grep "rows" $O/nb$nb.log; grep "bs_" $O/nb$nb/z_kernel_stats.csv | cut -d, -f1-4 | cut -c1-110
done
for nb in 5 4; do LFGPU_BS_V2_NB=$nb timeout -k 10 300 python3 tools/bench_lch.py 4096 16 4 2>&1 | grep rows; done
