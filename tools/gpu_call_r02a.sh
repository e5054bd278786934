set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02a
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a/pytest_gpu.log
tail -5 gpurun_out/r02a/pytest_gpu.log
timeout -k 10 120 tools/ubench > gpurun_out/r02a/ubench_int_rates.txt 2>&1
cat gpurun_out/r02a/ubench_int_rates.txt
for m in plain coop coop_reset hostpoll hostpoll_nofree; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats -d gpurun_out/r02a/repro_$m -o z -- tools/coop_exit_repro $m gpurun_out/r02a/maps_$m.txt > gpurun_out/r02a/repro_$m.log 2>&1
  echo "repro $m rc=$?" | tee -a gpurun_out/r02a/repro_summary.txt
done
LFGPU_SC_MODE=grid timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r02a/zk_grid -o z -- python3 tools/bench_zk.py 1 1 --no-cpu --maps=gpurun_out/r02a/maps_zk_grid.txt > gpurun_out/r02a/zk_grid.log 2>&1
echo "zk grid rc=$?" | tee -a gpurun_out/r02a/repro_summary.txt
LFGPU_SC_MODE=grid timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r02a/zk_grid_sd -o z -- python3 tools/bench_zk.py 1 1 --no-cpu --shutdown --maps=gpurun_out/r02a/maps_zk_grid_sd.txt > gpurun_out/r02a/zk_grid_sd.log 2>&1
echo "zk grid shutdown rc=$?" | tee -a gpurun_out/r02a/repro_summary.txt
cat gpurun_out/r02a/repro_summary.txt
