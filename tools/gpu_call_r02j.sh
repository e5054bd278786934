cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r02j
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_zk_cxx.py -m gpu -x -q -k "mdoc" > $O/pytest.log 2>&1; rc=$?; tail -15 $O/pytest.log; [ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/bench_zk.py 0 5 --mdoc > $O/zk_mdoc_hash.json 2> $O/zk_mdoc.err; echo "bench rc=$?"; tail -3 $O/zk_mdoc.err; cat $O/zk_mdoc_hash.json | cut -c1-1200
