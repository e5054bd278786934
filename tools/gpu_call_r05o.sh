#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_last.log 2>&1 || { tail -30 gpurun_out/pytest_gpu_last.log; exit 1; }
tail -2 gpurun_out/pytest_gpu_last.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 600 python bench.py --gpus 1 --steps 5 --warmup 2 > gpurun_out/bench_last.json 2> gpurun_out/bench_last.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_last.json").read().strip().splitlines()[-1])
print("K1", d["ms_per_step"], d["roofline"]["frac"], "flatsha32", d["zk_prove_flatsha256"]["total_ms"], "mdoc e2e", d["zk_prove_mdoc"].get("end_to_end", {}).get("prove_ms"))
PY
