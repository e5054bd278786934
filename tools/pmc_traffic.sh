#!/bin/bash
# Counter passes for this round's roofline.traffic figures (run on the GPU box from the repo root):
#   tools/pmc_traffic.sh OUTDIR
# One rocprofv3 --pmc pass per counter and program (TCC: FETCH_SIZE and WRITE_SIZE do not fit one pass; no trace domains next
# to --pmc; the program itself directly after "--").  tools/pmc_fold.py turns OUTDIR into profiles/r03/pmc_traffic.json.
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
run() {  # name counter program args...
  local name=$1 ctr=$2; shift 2
  echo "pmc: $name $ctr" >&2
  rocprofv3 --pmc "$ctr" --output-format csv -d "$OUT/$name/$ctr" -- "$@" > "$OUT/$name.$ctr.log" 2>&1 || echo "pmc: $name $ctr FAILED" >&2
}
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU GRBM_GUI_ACTIVE; do
  run k1 $C python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-secondary
  run slig $C python3 tools/slig_probe.py "$OUT/slig_probe.$C.txt" 20 1024
  run lch $C python3 tools/bench_lch.py 1024 20 5
  run zk32 $C python3 tools/bench_zk.py 32 2
  run zksig $C python3 tools/bench_zk.py 1 2 --mdoc-sig
done
echo "pmc: done" >&2
