#!/bin/bash
# Round 5: after sending real interpolators and complex x8 to poly_rt_dma_kernel: parity, the product's times, and complex x5 ... x7 either way
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05q
mkdir -p $O
cd $R
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_general_rate.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -20 $O/pytest.log; echo pytest failed; exit 1; }
tail -2 $O/pytest.log
SHAPES="interpolate" timeout -k 10 300 python3 scripts/time_real_shapes.py > $O/real_interp.txt 2>&1 && cat $O/real_interp.txt
EXTRA=3 SHAPES="interpolate" timeout -k 10 400 python3 scripts/time_shapes.py > $O/cplx_interp.txt 2>&1 && cut -c1-100 $O/cplx_interp.txt
export SFE_RT_DMA_SP1=1 SFE_RT_DMA_FORCE=1
EXTRA=3 BARE=1 SHAPES="x5,x6,x7" timeout -k 10 300 python3 scripts/time_shapes.py > $O/cplx_567_dma.txt 2>&1 && cut -c1-100 $O/cplx_567_dma.txt
echo collected
