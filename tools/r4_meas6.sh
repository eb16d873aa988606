cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s -p no:cacheprovider > gpurun_out/r4_gputest2.log 2>&1; echo "gpu tests exit $?"; grep -h "trajectory\]\|passed\|failed\|^E " gpurun_out/r4_gputest2.log | tail -12
