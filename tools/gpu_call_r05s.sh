#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_gfx3.log 2>&1 || { tail -30 gpurun_out/pytest_gpu_gfx3.log; exit 1; }
tail -2 gpurun_out/pytest_gpu_gfx3.log
for spec in "1" "32" "1 --mdoc"; do
  timeout -k 10 300 python tools/bench_zk.py $spec 6 > /tmp/o.json 2> /tmp/e.txt || { tail -5 /tmp/e.txt; exit 1; }
  python - "$spec" <<'PY'
import json, sys
d = json.loads(open("/tmp/o.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["wire_bytes_identical_to_reference"], "sumcheck", d["gpu_cxx_driver_ms"]["sumcheck"], sorted(d["gpu_cxx_driver_total_ms_all_reps"]), "verify", d.get("gpu_verify_ms"))
PY
done
