#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_p256_gpu.py tests/test_zk_cxx.py tests/test_sumcheck_drivers.py -m gpu -x -q -k "p256 or P256 or sig or mdoc or 256" > gpurun_out/p256_asm_tests.log 2>&1 || { tail -30 gpurun_out/p256_asm_tests.log; exit 1; }
tail -2 gpurun_out/p256_asm_tests.log
timeout -k 10 300 python tools/bench_zk.py 1 5 --mdoc-sig > /tmp/o.json 2> /tmp/e.txt || { tail -5 /tmp/e.txt; exit 1; }
python - <<'PY'
import json
d = json.loads(open("/tmp/o.json").read().strip().splitlines()[-1])
print("mdoc sig", d["wire_bytes_identical_to_reference"], d["gpu_cxx_driver_ms"], sorted(d["gpu_cxx_driver_total_ms_all_reps"]))
PY
timeout -k 10 300 python tools/stress_zk256.py 300 2>&1 | tail -1
timeout -k 5 120 ./tools/ubench_p256 2>&1 | tail -2
