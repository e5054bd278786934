cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r04e
mkdir -p $O
for cfg in "512 64" "1024 64" "2048 64" "256 128" "512 32" "1024 32" "512 128"; do
set -- $cfg
for w in "0 5 --mdoc" "32 5"; do
LFGPU_SC_PER_WG=$1 LFGPU_SC_WGS=$2 timeout -k 10 300 python tools/bench_zk.py $w --no-cpu > $O/zk.json 2>/dev/null || exit 1
python3 -c "
import json;d=json.load(open('$O/zk.json'));print('per_wg=$1 wgs=$2 [$w]', d['gpu_cxx_driver_ms']['sumcheck'], d['gpu_cxx_driver_ms']['wall_total'])"
done; done
