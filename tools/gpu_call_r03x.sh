cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r03x
mkdir -p $O
for cfg in "512 512" "1024 1024" "2048 2048" "1024 512" "256 512"; do
set -- $cfg
LFGPU_P256_GRID_MAX=$1 LFGPU_P256_PER_WG=$2 timeout -k 10 120 python tools/bench_zk.py 1 5 --mdoc-sig > $O/zk.json 2> $O/zk.err || exit 1
python3 -c "
import json;d=json.load(open('$O/zk.json'));print('grid_max=$1 per_wg=$2', d['gpu_cxx_driver_ms']['sumcheck'], d['gpu_cxx_driver_ms']['wall_total'], d['wire_bytes_identical_to_reference'])"
done
