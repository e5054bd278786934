cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
for i in 1 2; do
LFGPU_LIB=$GRAFT_REPO_ROOT/tools/liblfgpu_old.so timeout -k 10 300 python tools/bench_lch.py 1024 20 5 2>/dev/null | tail -1 | sed 's/^/old: /'
timeout -k 10 300 python tools/bench_lch.py 1024 20 5 2>/dev/null | tail -1 | sed 's/^/new: /'
done
