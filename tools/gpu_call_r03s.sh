cd $GRAFT_REPO_ROOT
ulimit -c 0
timeout -k 10 60 tools/ubench_p256
