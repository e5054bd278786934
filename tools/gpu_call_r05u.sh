#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_zk_cxx.py tests/test_sumcheck_drivers.py -m gpu -x -q -k "fp or FP or Fp" > gpurun_out/fp128_sgpr_tests.log 2>&1 || { tail -30 gpurun_out/fp128_sgpr_tests.log; exit 1; }
tail -2 gpurun_out/fp128_sgpr_tests.log
timeout -k 10 600 python bench.py --no-cpu-baseline --no-secondary --steps 20 > gpurun_out/bench_k1.json 2> gpurun_out/bench_k1.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_k1.json").read().strip().splitlines()[-1])
print("K1", d["ms_per_step"], d["roofline"]["frac"])
PY
timeout -k 10 300 python tools/bench_zk.py 1 5 --fp128 > /tmp/o.json 2> /tmp/e.txt
python - <<'PY'
import json
d = json.loads(open("/tmp/o.json").read().strip().splitlines()[-1])
print("flatsha fp128 nb1", d["wire_bytes_identical_to_reference"], sorted(d["gpu_cxx_driver_total_ms_all_reps"]))
PY
