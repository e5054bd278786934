cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r03d
mkdir -p $O
T0=$(date +%s); timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err; rc=$?; echo "bench wall $(( $(date +%s) - T0 )) s"; [ $rc -eq 0 ] || { tail -5 $O/bench.err; exit 1; }
python3 - <<PY
import json
d=json.load(open("$O/bench.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["gf2128_lch14_fft"]["ms_per_step"], d["ligero_commit_slig"]["rs_encode_ms"], d["zk_prove_flatsha256"]["total_ms"])
print(json.dumps(d["zk_prove_mdoc"].get("end_to_end")), d["zk_prove_mdoc"]["total_ms"])
PY
