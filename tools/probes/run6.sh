cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_network.py -x -q -k "two_stream" 2>&1 | grep -v amdgpu.ids | tail -3
bash tools/ab_env.sh BUDDY_OVERLAP 0 1 2 2>&1 | grep -v amdgpu.ids
bash tools/overlap_trace.sh 1 2>&1 | grep -v amdgpu.ids
