#!/bin/bash
# single-wave tail of the resident sumcheck kernel: parity (wire bytes), then timings with and without it
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_sumcheck_drivers.py tests/test_sumcheck_e2e.py tests/test_sumcheck_layer_random.py tests/test_zk_cxx.py -m gpu -x -q > gpurun_out/wave_tail_tests.log 2>&1 || { tail -30 gpurun_out/wave_tail_tests.log; exit 1; }
tail -3 gpurun_out/wave_tail_tests.log
for nb in 1 32; do
  for wt in 0 1; do
    LFGPU_SC_WAVE_TAIL=$wt timeout -k 10 300 python tools/bench_zk.py $nb 5 > gpurun_out/zk_nb${nb}_wt${wt}.json 2> gpurun_out/zk_nb${nb}_wt${wt}.err
    python - <<PY
import json
d = json.loads(open("gpurun_out/zk_nb${nb}_wt${wt}.json").read().strip().splitlines()[-1])
print("nb", $nb, "wave_tail", $wt, d["wire_bytes_identical_to_reference"], d["gpu_cxx_driver_ms"]["sumcheck"], d["gpu_cxx_driver_total_ms_all_reps"])
PY
  done
done
