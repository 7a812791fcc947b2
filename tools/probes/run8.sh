cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_kernels.py -x -q -k "flash_attention" 2>&1 | grep -v amdgpu.ids | tail -3
python -m pytest tests/test_hip_network.py -x -q -k "attention" 2>&1 | grep -v amdgpu.ids | tail -3
python -m pytest tests/test_hip_fullsize.py -x -q -k "long or f16 or 480000" 2>&1 | grep -v amdgpu.ids | tail -3
bash tools/prof_longform.sh r05b f16 2>&1 | grep -v amdgpu.ids | tail -18
python -c "
import json; d=json.load(open('gpurun_out/r05b_longform_f16_under_rocprof.json')); print('f16 longform ms/step (under rocprof)', d['ms_per_step'])"
python bench.py --length 480000 --batch 4 --steps 3 --warmup 1 --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest --attention f16 2>/dev/null > gpurun_out/r05b_longform_f16.json; python -c "
import json; d=json.load(open('gpurun_out/r05b_longform_f16.json')); print('f16 longform ms/step', d['ms_per_step'])"
