#!/bin/bash
# Round 5: the real-stream FIR with paired lanes (8-byte requests, v_permlane16_swap): parity, then the BASELINE workloads on real streams
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05u
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_variants.py tests/test_gpu_dropin.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; echo pytest failed; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 300 python3 scripts/time_real_baseline.py > $O/baseline_real.txt 2>&1 && cat $O/baseline_real.txt
echo collected
