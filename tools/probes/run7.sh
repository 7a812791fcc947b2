cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_network.py tests/test_hip_kernels.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
python -m pytest tests/test_hip_cli.py -x -q -k "bench_line" -s 2>&1 | grep -v amdgpu.ids | tail -5
bash tools/sweep_sub_batches.sh 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_sub_batch_sweep.txt
