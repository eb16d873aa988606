cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r4_gputest3.log 2>&1; echo "gpu tests exit $?"; tail -3 gpurun_out/r4_gputest3.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
