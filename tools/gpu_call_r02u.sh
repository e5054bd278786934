cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r02u
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_zk_cxx.py -m gpu -x -q -s -k "mdoc_sig" > $O/pytest_sig.log 2>&1; rc=$?; tail -6 $O/pytest_sig.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/bench_zk.py 1 5 --mdoc-sig > $O/zk_mdoc_sig.json 2> $O/zk_mdoc_sig.err; rc=$?; tail -3 $O/zk_mdoc_sig.err; cat $O/zk_mdoc_sig.json; [ $rc -eq 0 ] || exit 1
