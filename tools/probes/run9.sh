cd $GRAFT_REPO_ROOT
for sh in "29696 128 128" "7680 256 256"; do
  for d in 0 1 0 1; do BUDDY_WGEMM_PRIO=$d python tools/wgemm_one.py $sh 64 5 2>&1 | grep wgemm | sed "s/^/PRIO=$d /"; done
done
bash tools/ab_env.sh BUDDY_WGEMM_PRIO 0 1 2
