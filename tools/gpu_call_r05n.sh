#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_sumcheck_drivers.py tests/test_sumcheck_e2e.py tests/test_sumcheck_layer_random.py tests/test_zk_cxx.py -m gpu -x -q > gpurun_out/partials_post_tests.log 2>&1 || { tail -30 gpurun_out/partials_post_tests.log; exit 1; }
tail -2 gpurun_out/partials_post_tests.log
for spec in "1 --mdoc" "32"; do
  timeout -k 10 300 python tools/bench_zk.py $spec 6 > /tmp/o.json 2> /tmp/e.txt || { tail -5 /tmp/e.txt; exit 1; }
  python - "$spec" <<'PY'
import json, sys
d = json.loads(open("/tmp/o.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["wire_bytes_identical_to_reference"], "sumcheck", d["gpu_cxx_driver_ms"]["sumcheck"], sorted(d["gpu_cxx_driver_total_ms_all_reps"]))
PY
done
