cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r03l
mkdir -p $O
for gm in 65536 131072 262144; do
for cfg in "32 5" "0 5 --mdoc"; do
LFGPU_SC_GRID_MAX=$gm timeout -k 10 300 python tools/bench_zk.py $cfg --no-cpu > $O/zk.json 2>/dev/null || exit 1
python3 -c "
import json;d=json.load(open('$O/zk.json'));print('grid_max=$gm', '$cfg', d['gpu_cxx_driver_ms']['sumcheck'], d['gpu_cxx_driver_ms']['wall_total'])"
done; done
