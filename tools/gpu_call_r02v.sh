cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r02v
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_zk_cxx.py -m gpu -x -q -s -k "mdoc_sig" > $O/pytest_sig.log 2>&1; rc=$?; tail -6 $O/pytest_sig.log; [ $rc -eq 0 ] || exit 1
for sm in 0 1024 2048 4096 8192 16384; do
LFGPU_P256_SMALL=$sm timeout -k 10 300 python tools/bench_zk.py 1 5 --mdoc-sig > $O/zk_mdoc_sig_$sm.json 2> $O/zk_mdoc_sig.err || exit 1
python3 -c "
import json;d=json.load(open('$O/zk_mdoc_sig_$sm.json'));print($sm, d['gpu_cxx_driver_ms']['sumcheck'], d['gpu_cxx_driver_ms']['wall_total'], d['wire_bytes_identical_to_reference'])"
done
