cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r03o
mkdir -p $O
timeout -k 10 120 python -m pytest tests/test_zk_cxx.py -m gpu -x -q -s -k "small_p256" > $O/pytest_small.log 2>&1; rc=$?; tail -6 $O/pytest_small.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 120 python -m pytest tests/test_zk_cxx.py -m gpu -x -q -s -k "mdoc_sig" > $O/pytest_sig.log 2>&1; rc=$?; tail -6 $O/pytest_sig.log; [ $rc -eq 0 ] || exit 1
for gsel in 1 0; do
LFGPU_P256_GRID=$gsel timeout -k 10 120 python tools/bench_zk.py 1 5 --mdoc-sig > $O/zk_mdoc_sig_$gsel.json 2> $O/zk_mdoc_sig.err || exit 1
python3 -c "
import json;d=json.load(open('$O/zk_mdoc_sig_$gsel.json'));print('grid=$gsel', d['gpu_cxx_driver_ms']['sumcheck'], d['gpu_cxx_driver_ms']['wall_total'], d['wire_bytes_identical_to_reference'])"
done
