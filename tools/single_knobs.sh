cd $GRAFT_REPO_ROOT
for n in 16384 8192; do
  d=16; [ $n = 8192 ] && d=8
  F="python3 tools/shard_profile.py --single --n $n --d $d --reps 4"
  $F || exit 1
  for cr in 40 160; do $F --opt chain_rows=$cr || exit 1; done
  for lr in 96 512 1024; do $F --opt link_rows=$lr || exit 1; done
  $F --opt strip_min=1024 || exit 1
  $F --opt strip_min=256 || exit 1
done
