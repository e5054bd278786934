#!/bin/bash
# final measurement set of the round, all on one box: whole GPU suite, smoke, the bench line, its rocprofv3 kernel stats,
# the ZK driver on every fixture
set -e
export TMPDIR=/tmp
O=gpurun_out/final
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o z -- python3 bench.py --no-cpu-baseline > $O/prof_bench.json 2> $O/prof_bench.err; echo "rocprof rc=$?"
cp $O/prof_bench/*/z_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null || cp $O/prof_bench/z_kernel_stats.csv $O/bench_kernel_stats.csv
for nb in 1 2 4 8 16 32 33; do
  timeout -k 10 300 python tools/bench_zk.py $nb 5 > $O/zk_cxx_flatsha_nb$nb.json 2> $O/zk.err
done
timeout -k 10 300 python tools/bench_zk.py 1 5 --fp128 > $O/zk_cxx_flatsha_fp128_nb1.json 2> $O/zk.err
timeout -k 10 300 python tools/bench_zk.py 1 5 --mdoc > $O/zk_cxx_mdoc_hash.json 2> $O/zk.err
timeout -k 10 300 python tools/bench_zk.py 1 5 --mdoc-sig > $O/zk_cxx_mdoc_sig.json 2> $O/zk.err
timeout -k 10 600 ./oracle/_ref/mdoc_gpu 5 > $O/mdoc_end_to_end.log 2> $O/mdoc_e2e.err || { tail -5 $O/mdoc_e2e.err; exit 1; }
find $O -name "*.csv" -size +20M -delete
rm -rf $O/prof_bench
python - <<'PY'
import json, glob
O = "gpurun_out/final"
d = json.loads(open(O + "/bench.json").read().strip().splitlines()[-1])
print("K1", d["ms_per_step"], d["value"], d["roofline"]["frac"], "lch", d["gf2128_lch14_fft"]["ms_per_step"], "f64", d["f64_2_fft"]["ms_per_step"])
print("slig", {k: v for k, v in d["ligero_commit_slig"].items() if "ms" in k}, "cpu", d["cpu_baseline"]["value"])
for f in sorted(glob.glob(O + "/zk_cxx_*.json")):
    z = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], z["wire_bytes_identical_to_reference"], min(z["gpu_cxx_driver_total_ms_all_reps"]), z.get("gpu_verify_ms"), z["cpu_reference_ms"].get("total"))
print(open(O + "/mdoc_end_to_end.log").read().strip().splitlines()[-1][:700])
PY
