cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r03q
mkdir -p $O
for gm in 1024 2048 4096 8192 16384 32768; do
for pw in 256 512; do
LFGPU_P256_GRID_MAX=$gm LFGPU_P256_PER_WG=$pw timeout -k 10 120 python tools/bench_zk.py 1 5 --mdoc-sig > $O/zk.json 2> $O/zk.err || exit 1
python3 -c "
import json;d=json.load(open('$O/zk.json'));print('grid_max=$gm per_wg=$pw', d['gpu_cxx_driver_ms']['sumcheck'], d['gpu_cxx_driver_ms']['wall_total'], d['wire_bytes_identical_to_reference'])"
done; done
