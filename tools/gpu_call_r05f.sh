#!/bin/bash
# single-wave tail of the P-256 grid kernel: parity (mdoc signature circuit, small P-256 fixture, reference integration), then timings
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_zk_cxx.py tests/test_p256_gpu.py -m gpu -x -q > gpurun_out/wave_tail256_tests.log 2>&1 || { tail -30 gpurun_out/wave_tail256_tests.log; exit 1; }
tail -3 gpurun_out/wave_tail256_tests.log
for wt in 0 1; do
  LFGPU_P256_WAVE_TAIL=$wt timeout -k 10 300 python tools/bench_zk.py 1 5 --mdoc-sig > gpurun_out/zk_sig_wt${wt}.json 2> gpurun_out/zk_sig_wt${wt}.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/zk_sig_wt${wt}.json").read().strip().splitlines()[-1])
print("mdoc sig wave_tail", $wt, d["wire_bytes_identical_to_reference"], d["gpu_cxx_driver_ms"], d["gpu_cxx_driver_total_ms_all_reps"])
PY
done
timeout -k 10 600 python tools/stress_zk256.py 60 > gpurun_out/stress256_wt.log 2>&1 || { tail -5 gpurun_out/stress256_wt.log; exit 1; }
tail -2 gpurun_out/stress256_wt.log
