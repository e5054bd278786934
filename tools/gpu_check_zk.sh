# GPU box: parity tests of the sumcheck / ZK drivers, then single-proof latency and K = 16 throughput with the A/B switches
# of the dispatch-saving paths (LFGPU_EQ_FUSED=0: EQ factor tables and their combination in two launches; LFGPU_SC_BIND_SPLIT=1:
# Dense::bind and HQuad::bind_h in two launches).  Usage: gpurun -- 'bash tools/gpu_check_zk.sh [outdir]'
set -e
O=${1:-gpurun_out/zkcheck}
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_sumcheck_drivers.py tests/test_sumcheck_e2e.py tests/test_sumcheck_layer_random.py tests/test_zk_cxx.py tests/test_zk_concurrent.py tests/test_reference_kats.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for nb in 1 32; do
  python tools/bench_zk.py $nb 7 > $O/zk$nb.json 2> $O/zk$nb.err
  LFGPU_EQ_FUSED=0 LFGPU_SC_BIND_SPLIT=1 python tools/bench_zk.py $nb 7 > $O/zk${nb}_split.json 2>> $O/zk$nb.err
done
for rep in 1 2 3; do
  python tools/zk_throughput.py --jobs flatsha32 --k 16 --seconds 4 --json > $O/thr_new_$rep.json 2>> $O/thr.err
  LFGPU_EQ_FUSED=0 python tools/zk_throughput.py --jobs flatsha32 --k 16 --seconds 4 --json > $O/thr_eqsplit_$rep.json 2>> $O/thr.err
  LFGPU_SC_BIND_SPLIT=1 python tools/zk_throughput.py --jobs flatsha32 --k 16 --seconds 4 --json > $O/thr_bindsplit_$rep.json 2>> $O/thr.err
  LFGPU_EQ_FUSED=0 LFGPU_SC_BIND_SPLIT=1 python tools/zk_throughput.py --jobs flatsha32 --k 16 --seconds 4 --json > $O/thr_allsplit_$rep.json 2>> $O/thr.err
done
python tools/zk_throughput.py --jobs mdoc --k 16 --seconds 4 --json > $O/thr_mdoc.json 2>> $O/thr.err
python - $O <<'P'
import json,glob,sys
O=sys.argv[1]
for f in sorted(glob.glob(O+'/zk*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], d.get('gpu_cxx_driver_total_ms_all_reps'))
for f in sorted(glob.glob(O+'/thr_*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], {j:{k:round(v['proofs_per_s'],1) for k,v in d[j]['k'].items()} for j in d if isinstance(d[j],dict)})
P
