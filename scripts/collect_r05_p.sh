#!/bin/bash
# Round 5 experiment: the pure interpolators (SP = 1) through the LDS-DMA kernel (diagnostic switches) against poly_rt1_kernel, complex and real
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05p
mkdir -p $O
cd $R
for V in 0 1; do
  if [ $V = 1 ]; then export SFE_RT_DMA_SP1=1 SFE_RT_DMA_FORCE=1; else unset SFE_RT_DMA_SP1 SFE_RT_DMA_FORCE; fi
  echo "== SP1 through poly_rt_dma_kernel: $V" >> $O/interpolators.txt
  BARE=1 SHAPES="interpolate" timeout -k 10 400 python3 scripts/time_shapes.py >> $O/interpolators.txt 2>&1 || echo cplx failed
  DIAG=1 SHAPES="interpolate" timeout -k 10 300 python3 scripts/time_real_shapes.py >> $O/interpolators.txt 2>&1 || echo real failed
done
cut -c1-110 $O/interpolators.txt
echo collected
