cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r02t
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -6 $O/pytest_gpu.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; rc=$?; tail -2 $O/bench.err; [ $rc -eq 0 ] || exit 1
python3 - <<PY
import json
d=json.load(open("$O/bench.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["gf2128_lch14_fft"]["ms_per_step"], d["zk_prove_flatsha256"]["total_ms"])
print(json.dumps(d["zk_prove_mdoc"])[:1500])
PY
