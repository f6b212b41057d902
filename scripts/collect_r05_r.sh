#!/bin/bash
# Round 5: the register-window kernel for real interpolators (poly_int4_dma_kernel): parity, then times against the one-sample-per-read form (SFE_RT_DMA_WINDOW=0)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05r
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "window or lds_dma" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; echo pytest failed; exit 1; }
tail -2 $O/pytest.log
for V in 1 0; do
  echo "== SFE_RT_DMA_WINDOW=$V" >> $O/real_interp.txt
  SFE_RT_DMA_WINDOW=$V DIAG=1 SHAPES="interpolate" timeout -k 10 300 python3 scripts/time_real_shapes.py >> $O/real_interp.txt 2>&1 || echo failed
done
cat $O/real_interp.txt
echo collected
