cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r03g
mkdir -p $O
LFGPU_LIB=$GRAFT_REPO_ROOT/tools/liblfgpu_old.so timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/old -o z -- python3 tools/bench_lch.py 1024 20 5 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/new -o z -- python3 tools/bench_lch.py 1024 20 5 > /dev/null 2>&1
for v in old new; do echo $v; grep -E "bs_" $O/$v/z_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120; done
