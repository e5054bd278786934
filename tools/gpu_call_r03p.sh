cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r03p
mkdir -p $O
for pw in 256 512 1024 2048 4096 8192; do
LFGPU_P256_PER_WG=$pw timeout -k 10 120 python tools/bench_zk.py 1 5 --mdoc-sig > $O/zk_$pw.json 2> $O/zk.err || exit 1
python3 -c "
import json;d=json.load(open('$O/zk_$pw.json'));print('per_wg=$pw', d['gpu_cxx_driver_ms']['sumcheck'], d['gpu_cxx_driver_ms']['wall_total'], d['wire_bytes_identical_to_reference'])"
done
