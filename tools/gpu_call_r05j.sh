#!/bin/bash
# sweep of the resident-grid knobs on the mdoc hash circuit (GF2_128, 17 layers up to 2^20 wide)
mkdir -p gpurun_out
: > gpurun_out/sweep_mdoc_hash.txt
run() {
  env "$@" timeout -k 10 200 python tools/bench_zk.py 1 4 --mdoc > /tmp/o.json 2> /tmp/e.txt || { echo "FAIL $*" >> gpurun_out/sweep_mdoc_hash.txt; return; }
  python - "$*" <<'PY' >> gpurun_out/sweep_mdoc_hash.txt
import json, sys
d = json.loads(open("/tmp/o.json").read().strip().splitlines()[-1])
print(sys.argv[1], "identical", d["wire_bytes_identical_to_reference"], "sumcheck", d["gpu_cxx_driver_ms"]["sumcheck"], "total", min(d["gpu_cxx_driver_total_ms_all_reps"]))
PY
}
run X=1
run LFGPU_SC_PER_WG=256
run LFGPU_SC_PER_WG=1024
run LFGPU_SC_PER_WG=256 LFGPU_SC_WGS=128
run LFGPU_SC_WGS=128
run LFGPU_SC_WGS=32
run LFGPU_SC_GRID_MAX=65536
run LFGPU_SC_GRID_MAX=32768
run LFGPU_SC_GRID_MAX=16384
run LFGPU_SC_GRID_MAX=16384 LFGPU_SC_PER_WG=256
run X=2
cat gpurun_out/sweep_mdoc_hash.txt
