#!/bin/bash
# Round 5, seventh GPU call: the virtual-memory probe with the time course (G, H), and long decimations at several tile sizes
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05g
mkdir -p $O
cd $R
timeout -k 10 200 scripts/probes/vmm_loss 3 > $O/vmm_loss.txt 2>&1 || echo vmm probe failed
cat $O/vmm_loss.txt
for TM in default 256 128 64; do
  if [ $TM = default ]; then unset SFE_RT_TM; else export SFE_RT_TM=$TM; fi
  echo "== SFE_RT_TM=$TM" >> $O/shapes_long_decimations.txt
  EXTRA=1 SHAPES="by 12,by 13,by 16,by 24,by 32,by 48" timeout -k 10 300 python3 scripts/time_shapes.py >> $O/shapes_long_decimations.txt 2>&1 || echo failed $TM
done
cut -c1-110 $O/shapes_long_decimations.txt
echo collected
