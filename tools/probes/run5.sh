cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_network.py -x -q -k "two_stream or fused_round4 or forward_vjp_vs_golden" 2>&1 | grep -v amdgpu.ids | tail -5
bash tools/ab_env.sh BUDDY_OVERLAP 0 1 2 2>&1 | grep -v amdgpu.ids
bash tools/probes/run4.sh
