cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r02d
mkdir -p $O
timeout -k 10 300 python tools/slig_probe.py $O/probe.txt 20 64,1024 2> $O/probe.err; rc=$?; echo "probe rc=$rc"; tail -3 $O/probe.err; [ $rc -eq 0 ] || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lch14 or rs_encode" > $O/pytest.log 2>&1; rc=$?; tail -5 $O/pytest.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 600 python bench.py --steps 5 --warmup 1 > $O/bench.json 2> $O/bench.err; rc=$?; tail -3 $O/bench.err; cat $O/bench.json | cut -c1-7000; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --force-dist --no-cpu-baseline > $O/bench_dist1.json 2> $O/bench_dist1.err; echo "force-dist rc=$?"; tail -3 $O/bench_dist1.err; python3 -c "
import json;d=json.load(open('$O/bench_dist1.json'));print(d.get('ligero_commit_sharded'))"
for nb in 1 2 4 8 16 32 33; do timeout -k 10 120 python tools/bench_zk.py $nb 5 > $O/zk_nb$nb.json 2>/dev/null; python3 -c "
import json;d=json.load(open('$O/zk_nb$nb.json'));print($nb, d['gpu_cxx_driver_ms']['wall_total'], d.get('cpu_reference_ms',{}).get('total'), d['gpu_verify_ms'])"; done
