cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for sh in "29696 128 128" "29696 128 256" "7680 256 256" "7680 256 512"; do
  for d in 0 1 0 1; do BUDDY_WGEMM_DMA=$d python tools/wgemm_one.py $sh 64 5 2>&1 | grep wgemm | sed "s/^/DMA=$d /"; done
done
bash tools/ab_env.sh BUDDY_WGEMM_DMA 0 1 2
bash tools/sweep_sub_batches.sh
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_run4.log
