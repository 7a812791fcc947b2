cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_network.py -x -q -k "architecture_family" 2>&1 | tail -12
python -m pytest tests/test_hip_multirank.py -x -q -s -k "eight" 2>&1 | tail -12
