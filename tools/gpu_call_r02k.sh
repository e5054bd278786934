cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r02k
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "field_binops or fp128 or fp_ or zk_over_fp128 or sumcheck or raw_eq2 or ligero or sharded or integration" > $O/pytest.log 2>&1; rc=$?; tail -6 $O/pytest.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-secondary --no-cpu-baseline > $O/bench.json 2> $O/bench.err; rc=$?; tail -2 $O/bench.err; python3 -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
timeout -k 10 120 tools/ubench 2>&1 | grep "fp128"
