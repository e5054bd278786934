cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r02z
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rs_encode or lch14 or ligero" > $O/pytest_rs.log 2>&1; rc=$?; tail -8 $O/pytest_rs.log; [ $rc -eq 0 ] || exit 1
for t in 1 0; do
LFGPU_BS_NW_MATCH=$t timeout -k 10 600 python bench.py --no-cpu-baseline > $O/bench_t$t.json 2> $O/bench_t$t.err || exit 1
python3 -c "
import json;d=json.load(open('$O/bench_t$t.json'));s=d['ligero_commit_slig'];print('nw_match=$t', s['rs_encode_ms'], s['checked_vs_oracle'], d['gf2128_lch14_fft']['ms_per_step'], d['ligero_commit_flatsha32'].get('rs_encode_ms'))"
done
