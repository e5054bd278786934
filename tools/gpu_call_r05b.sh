#!/bin/bash
# F64_2 FFT: parity tests, then the bench line (K1 must not have moved)
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "f64 or fp128_fft" > gpurun_out/f64_tests.log 2>&1 || { tail -30 gpurun_out/f64_tests.log; exit 1; }
tail -3 gpurun_out/f64_tests.log
timeout -k 10 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_f64.json 2> gpurun_out/bench_f64.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_f64.json").read().strip().splitlines()[-1])
print("K1 ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
print("f64_2", d.get("f64_2_fft"))
print("lch14", d["gf2128_lch14_fft"]["ms_per_step"])
PY
