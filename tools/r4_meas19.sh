cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r4_gputest5.log 2>&1; echo "gpu tests exit $?"; tail -12 gpurun_out/r4_gputest5.log
