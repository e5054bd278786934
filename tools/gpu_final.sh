# GPU box: the end-of-round measurement set.  Usage: gpurun --timeout 1200 -- 'bash tools/gpu_final.sh [outdir]'
set -e
O=${1:-gpurun_out/final}
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1 || { tail -40 $O/gputests.log; exit 1; }
tail -2 $O/gputests.log
python bench.py > $O/bench.json 2> $O/bench.err
python - $O <<'P'
import json,sys
d=json.loads(open(sys.argv[1]+'/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'])
z=d['zk_prove_flatsha256']; print({k:round(v['total_ms'],2) for k,v in z['by_sha_blocks'].items()})
print(d['zk_prove_mdoc'].get('total_ms'), {k:v for k,v in d['zk_prove_mdoc'].get('end_to_end',{}).items() if 'ms' in k})
print(d['zk_throughput']['proofs_per_s'])
print(d['gf2128_lch14_fft']['ms_per_step'], d['ligero_commit_slig'].get('rs_encode_ms'), d['cpu_baseline'])
P
