cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r04a
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_zk_cxx.py tests/test_reference_integration.py -m gpu -x -q -k "p256 or mdoc" > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 120 python tools/bench_zk.py 1 5 --mdoc-sig > $O/zk_mdoc_sig.json 2> $O/zk.err || exit 1
python3 -c "
import json;d=json.load(open('$O/zk_mdoc_sig.json'));print(d['gpu_cxx_driver_ms'])"
