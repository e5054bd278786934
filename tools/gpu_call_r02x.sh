cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r02x
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rs_encode or lch14 or ligero" > $O/pytest_rs.log 2>&1; rc=$?; tail -8 $O/pytest_rs.log; [ $rc -eq 0 ] || exit 1
for t in 1 0; do
LFGPU_RS_TOWER=$t timeout -k 10 600 python bench.py --no-cpu-baseline > $O/bench_t$t.json 2> $O/bench_t$t.err || exit 1
python3 -c "
import json;d=json.load(open('$O/bench_t$t.json'));s=d['ligero_commit_slig'];print('tower=$t', s['rs_encode_ms'], s['column_commit_ms'], s.get('oracle_check'), d['gf2128_lch14_fft']['ms_per_step'])"
done
