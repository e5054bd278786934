#!/bin/bash
mkdir -p gpurun_out
: > gpurun_out/sweep2.txt
run() {
  spec="$1"; shift
  env "$@" timeout -k 10 200 python tools/bench_zk.py $spec 5 > /tmp/o.json 2> /tmp/e.txt || { echo "FAIL $spec $*" >> gpurun_out/sweep2.txt; return; }
  python - "$spec $*" <<'PY' >> gpurun_out/sweep2.txt
import json, sys
d = json.loads(open("/tmp/o.json").read().strip().splitlines()[-1])
print(sys.argv[1], "identical", d["wire_bytes_identical_to_reference"], "sumcheck", d["gpu_cxx_driver_ms"]["sumcheck"], "total", min(d["gpu_cxx_driver_total_ms_all_reps"]))
PY
}
for spec in "1 --mdoc" "32"; do
run "$spec" X=1
run "$spec" LFGPU_SC_GRID_MAX=262144
run "$spec" LFGPU_SC_GRID_MAX=262144 LFGPU_SC_WGS=128
run "$spec" LFGPU_SC_WGS=128
run "$spec" LFGPU_SC_WGS=96
done
cat gpurun_out/sweep2.txt
