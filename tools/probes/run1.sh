cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_kernels.py -x -q -k "flash_attention" 2>&1 | tail -5
python -m pytest tests/test_hip_network.py -x -q -k "attention" 2>&1 | tail -5
python -m pytest tests/test_hip_fullsize.py -x -q 2>&1 | tail -5
python bench.py --length 480000 --batch 4 --steps 3 --warmup 1 --no-cpu-baseline --also-concurrent 0 --legs none --no-rccl-selftest --attention f16 > gpurun_out/r05a_longform_f16.json 2> gpurun_out/r05a_longform_f16.err; tail -3 gpurun_out/r05a_longform_f16.err
python -c "
import json; d=json.load(open('gpurun_out/r05a_longform_f16.json')); print('f16 longform ms/step', d['ms_per_step'])"
