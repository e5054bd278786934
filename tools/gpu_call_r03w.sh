cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r03w
mkdir -p $O
timeout -k 10 120 python -m pytest tests/test_zk_cxx.py -m gpu -x -q -k "small_p256 or mdoc_sig" > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -eq 0 ] || exit 1
for gm in 512 2048 8192 131072; do
LFGPU_P256_GRID_MAX=$gm timeout -k 10 120 python tools/bench_zk.py 1 5 --mdoc-sig > $O/zk.json 2> $O/zk.err || exit 1
python3 -c "
import json;d=json.load(open('$O/zk.json'));print('grid_max=$gm', d['gpu_cxx_driver_ms']['sumcheck'], d['gpu_cxx_driver_ms']['wall_total'], d['wire_bytes_identical_to_reference'])"
done
