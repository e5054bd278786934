#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_sha.log 2>&1 || { tail -30 gpurun_out/pytest_gpu_sha.log; exit 1; }
tail -2 gpurun_out/pytest_gpu_sha.log
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/bench_sha.json 2> gpurun_out/bench_sha.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_sha.json").read().strip().splitlines()[-1])
print("K1", d["ms_per_step"], "slig", d["ligero_commit_slig"])
print("flatsha32 commit", d["ligero_commit_flatsha32"])
z = d["zk_prove_flatsha256"]; print("zk32", z["commit_ms"], z["prove_ms"], z["total_ms"])
PY
