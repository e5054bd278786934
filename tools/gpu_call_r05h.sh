#!/bin/bash
# whole GPU suite, then the bench line
set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r05h.log 2>&1 || { tail -30 gpurun_out/pytest_gpu_r05h.log; exit 1; }
tail -3 gpurun_out/pytest_gpu_r05h.log
timeout -k 10 600 python bench.py > gpurun_out/bench_r05h.json 2> gpurun_out/bench_r05h.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r05h.json").read().strip().splitlines()[-1])
print("K1", d["ms_per_step"], d["roofline"]["frac"], "lch", d["gf2128_lch14_fft"]["ms_per_step"], "f64", d["f64_2_fft"]["ms_per_step"])
print("slig", {k: v for k, v in d["ligero_commit_slig"].items() if "ms" in k})
z = d["zk_prove_flatsha256"]; print("flatsha32", z.get("total_ms"), z.get("wire_bytes_identical_to_reference"))
m = d["zk_prove_mdoc"]; print("mdoc", json.dumps(m)[:900])
PY
