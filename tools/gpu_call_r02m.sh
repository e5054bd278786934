cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r02m
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -6 $O/pytest_gpu.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; rc=$?; tail -2 $O/bench.err; [ $rc -eq 0 ] || exit 1
python3 - <<PY
import json
d=json.load(open("$O/bench.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["gf2128_lch14_fft"]["ms_per_step"], d["ligero_commit_slig"]["rs_encode_ms"], d["ligero_commit_slig"]["column_commit_ms"], d["zk_prove_flatsha256"]["total_ms"], d["cpu_baseline"])
PY
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o z -- python3 bench.py --no-cpu-baseline > $O/prof_bench.json 2> $O/prof_bench.err; echo "prof rc=$?"
head -8 $O/prof_bench/z_kernel_stats.csv | cut -c1-120
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -c1-8)
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/k1_$n -o p -- python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-secondary > $O/k1_$n.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/k2_$n -o p -- python3 tools/bench_lch.py 1024 20 5 > $O/k2_$n.log 2>&1 || exit 1
done
python3 tools/pmc_summary.py $O/pmc_k1.json fp_fft_tile $O/k1_FETCH_SI $O/k1_WRITE_SI $O/k1_SQ_WAVES $O/k1_GRBM_GUI > /dev/null
python3 tools/pmc_summary.py $O/pmc_k2.json bs_ $O/k2_FETCH_SI $O/k2_WRITE_SI $O/k2_SQ_WAVES $O/k2_GRBM_GUI > /dev/null
python3 -c "
import json
for f in ('$O/pmc_k1.json','$O/pmc_k2.json'):
    d=json.load(open(f))
    for k,v in d.items(): print(k, {x:round(y) for x,y in v.items() if x in ('FETCH_SIZE','WRITE_SIZE','SQ_INSTS_VALU','SQ_WAVE_CYCLES','GRBM_GUI_ACTIVE','mean_ns_under_pmc')})"
for nb in 1 32; do timeout -k 10 120 python tools/bench_zk.py $nb 5 > $O/zk_nb$nb.json 2>/dev/null; python3 -c "
import json;d=json.load(open('$O/zk_nb$nb.json'));print($nb, d['gpu_cxx_driver_ms']['wall_total'], d.get('cpu_reference_ms',{}).get('total'), d['gpu_verify_ms'])"; done
timeout -k 10 120 python tools/bench_zk.py 1 5 --fp128 > $O/zk_fp_nb1.json 2>/dev/null; python3 -c "
import json;d=json.load(open('$O/zk_fp_nb1.json'));print('fp128', d['gpu_cxx_driver_ms']['wall_total'], d.get('cpu_reference_ms',{}).get('total'), d['gpu_verify_ms'])"
find $O -name "*.csv" -size +20M -delete
