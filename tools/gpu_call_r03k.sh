cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r03k
mkdir -p $O
timeout -k 10 600 python tools/stress_zk256.py 300 > $O/stress_zk256.log 2>&1; rc=$?; tail -3 $O/stress_zk256.log; [ $rc -eq 0 ] || exit 1
