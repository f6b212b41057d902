#!/bin/bash
# Round 5: the register-window kernel at input steps 2 ... 5 on real streams: parity, then the compile-time shapes and the runtime ones with the
# window form limited to SP = 1 (SFE_RT_DMA_WINDOW=1: the tree before) and up to SP = 5 (=5), diagnostic library, one process each
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05x
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "window or lds_dma or integer_step" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; echo pytest failed; exit 1; }
tail -2 $O/pytest.log
for V in 1 5; do
  export SFE_RT_DMA_WINDOW=$V
  echo "== SFE_RT_DMA_WINDOW=$V" >> $O/window_real.txt
  DIAG=1 timeout -k 10 500 python3 scripts/time_real_compiled.py >> $O/window_real.txt 2>&1 || echo failed $V
  DIAG=1 SHAPES="4/5,2/5,3/5" timeout -k 10 300 python3 scripts/time_real_shapes.py >> $O/window_real.txt 2>&1 || echo failed shapes $V
done
cut -c1-60 $O/window_real.txt
echo collected
