#!/bin/bash
# Round 5: wire-format (u8) input through the LDS-DMA kernels: parity, then times against the kernels it ran before (SFE_RT_DMA_U8=0) and with the
# DMA kernels forced onto the compile-time shapes too (SFE_RT_DMA_FORCE=1); diagnostic library, one process each
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05z
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_dropin.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; echo pytest failed; exit 1; }
tail -2 $O/pytest.log
echo "== before (SFE_RT_DMA_U8=0)" >> $O/shapes_u8.txt
SFE_RT_DMA_U8=0 DIAG=1 timeout -k 10 400 python3 scripts/time_u8_shapes.py >> $O/shapes_u8.txt 2>&1 || echo failed 0
echo "== after (default dispatch)" >> $O/shapes_u8.txt
DIAG=1 timeout -k 10 400 python3 scripts/time_u8_shapes.py >> $O/shapes_u8.txt 2>&1 || echo failed 1
cat $O/shapes_u8.txt
echo collected
