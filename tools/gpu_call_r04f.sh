cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r04f
mkdir -p $O
timeout -k 10 600 python bench.py --gpus 1 --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; rc=$?; [ $rc -eq 0 ] || { tail -5 $O/bench.err; exit 1; }
python3 -c "
import json;d=json.load(open('$O/bench.json'));print(d['n_gpus'], d['steps'], d['warmup'], d['value'], d['roofline']['frac'], d['zk_prove_mdoc']['total_ms'], d['zk_prove_mdoc']['end_to_end']['prove_ms'])"
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --force-dist > $O/bench_dist.json 2> $O/bench_dist.err; rc=$?; [ $rc -eq 0 ] || { tail -5 $O/bench_dist.err; exit 1; }
python3 -c "
import json;d=json.load(open('$O/bench_dist.json'));print('dist', d['n_gpus'], d['value'], d.get('ligero_commit_sharded'))"
