#!/bin/bash
# Round 5, twelfth GPU call: the runtime-shape kernel for odd input steps with its tile fetched by LDS-DMA (poly_rt_dma.hip) against poly_rt_kernel,
# one process per setting (SFE_RT_DMA: diagnostic library), the same shapes; then the tests that cover those shapes.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05l
mkdir -p $O
cd $R
for D in 1 0; do
  export SFE_RT_DMA=$D
  echo "== SFE_RT_DMA=$D" >> $O/shapes_rt_dma.txt
  BARE=1 EXTRA=2 SHAPES="4/5,6/5,5/6,7/8,3/5,2/5,9/5,11/8,7/3,by 7" timeout -k 10 400 python3 scripts/time_shapes.py >> $O/shapes_rt_dma.txt 2>&1 || echo failed $D
done
unset SFE_RT_DMA
cut -c1-125 $O/shapes_rt_dma.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
echo collected
