cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r03y
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -3 $O/pytest_gpu.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/stress_zk256.py 300 > $O/stress.log 2>&1; rc=$?; tail -1 $O/stress.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 120 python tools/bench_zk.py 1 5 --mdoc-sig > $O/zk_mdoc_sig.json 2> $O/zk.err || exit 1
python3 -c "
import json;d=json.load(open('$O/zk_mdoc_sig.json'));print(d['gpu_cxx_driver_ms'])"
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || exit 1
python3 -c "
import json;d=json.load(open('$O/bench.json'));m=d['zk_prove_mdoc'];print(d['value'], d['gf2128_lch14_fft']['ms_per_step'], d['ligero_commit_slig']['rs_encode_ms'], m['hash']['total_ms'], m['sig']['total_ms'], m['total_ms'], m.get('end_to_end',{}).get('prove_ms'), m.get('end_to_end',{}).get('verify_ms'))"
