cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -x -k "skinny" 2>&1 | tail -15
