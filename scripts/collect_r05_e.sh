#!/bin/bash
# Round 5, fifth GPU call: do the shapes that mostly read run faster from smaller tiles (more resident workgroups)?  SFE_RT_TM (diagnostic library).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05e
mkdir -p $O
cd $R
for TM in default 512 256 128; do
  if [ $TM = default ]; then unset SFE_RT_TM; else export SFE_RT_TM=$TM; fi
  echo "== SFE_RT_TM=$TM" >> $O/shapes_tile_m.txt
  BARE=1 SHAPES="by 6,by 7,by 16,7/4,7/3,4/5,3/2" timeout -k 10 300 python3 scripts/time_shapes.py >> $O/shapes_tile_m.txt 2>&1 || echo failed $TM
done
cat $O/shapes_tile_m.txt
echo collected
