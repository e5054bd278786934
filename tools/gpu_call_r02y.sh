cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r02y
mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o z -- python3 tools/slig_probe.py $O/slig_steps.txt 20 1024 > $O/slig.log 2> $O/slig.err; echo "rc=$?"; tail -3 $O/slig.log
