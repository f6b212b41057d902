#!/bin/bash
# Round 5: the compile-time tiled shapes on real and complex streams, default dispatch against the LDS-DMA kernels forced onto them
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05w
mkdir -p $O
cd $R
for V in 0 1; do
  if [ $V = 1 ]; then export SFE_RT_DMA_FORCE=1; else unset SFE_RT_DMA_FORCE; fi
  DIAG=1 timeout -k 10 500 python3 scripts/time_real_compiled.py >> $O/compiled_real.txt 2>&1 || echo failed $V
done
cat $O/compiled_real.txt
echo collected
