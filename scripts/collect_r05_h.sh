#!/bin/bash
# Round 5, eighth GPU call: long decimations at several tile sizes (diagnostic library: SFE_RT_TM is honoured there only)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05h
mkdir -p $O
cd $R
for TM in default 256 128 64; do
  if [ $TM = default ]; then unset SFE_RT_TM; else export SFE_RT_TM=$TM; fi
  echo "== SFE_RT_TM=$TM" >> $O/shapes_long_decimations.txt
  BARE=1 EXTRA=1 SHAPES="by 12,by 13,by 16,by 24,by 32,by 48" timeout -k 10 300 python3 scripts/time_shapes.py >> $O/shapes_long_decimations.txt 2>&1 || echo failed $TM
done
cut -c1-110 $O/shapes_long_decimations.txt
echo collected
