#!/bin/bash
# Round 5, sixth GPU call: the shapes that mostly read with the tap loop unrolled deeper for one or two m per thread, at three tile sizes;
# and the virtual-memory probe (when does a fresh mapping lose what is written to it?)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05f
mkdir -p $O
cd $R
timeout -k 10 120 scripts/probes/vmm_loss 3 > $O/vmm_loss.txt 2>&1 || echo vmm probe failed
cat $O/vmm_loss.txt
for TM in default 256 128; do
  if [ $TM = default ]; then unset SFE_RT_TM; else export SFE_RT_TM=$TM; fi
  echo "== SFE_RT_TM=$TM" >> $O/shapes_tile_m.txt
  BARE=1 SHAPES="by 6,by 7,by 16,7/4,7/3,4/5" timeout -k 10 300 python3 scripts/time_shapes.py >> $O/shapes_tile_m.txt 2>&1 || echo failed $TM
done
cut -c1-110 $O/shapes_tile_m.txt
echo collected
