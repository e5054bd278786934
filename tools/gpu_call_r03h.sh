cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r03h
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rs_encode or lch14" > $O/pytest_rs.log 2>&1; rc=$?; tail -2 $O/pytest_rs.log; [ $rc -eq 0 ] || exit 1
for i in 1 2; do
LFGPU_LIB=$GRAFT_REPO_ROOT/tools/liblfgpu_old.so timeout -k 10 300 python tools/bench_lch.py 1024 20 5 2>/dev/null | tail -1 | sed 's/^/old: /'
timeout -k 10 300 python tools/bench_lch.py 1024 20 5 2>/dev/null | tail -1 | sed 's/^/new: /'
done
