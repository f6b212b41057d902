#!/bin/bash
# Round 5, second GPU call: the general-rate kernel after the branch-free shares loop -- its tests, its timing, its counters (one shape).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05b
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_general_rate.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" | tee -a $O/pytest.txt
tail -5 $O/pytest.txt
LOG2N=28 GENERAL_ONLY=1 timeout -k 10 300 python3 scripts/time_general_rate.py > $O/general_rate.txt 2>&1 || echo timing failed
cat $O/general_rate.txt
export LOG2N=28 GENERAL_ONLY=1 RATES=1.77 DEFAULT_ONLY=1
bash scripts/prof_pmc.sh r05b/sq_general_1p77 poly_gen -- python3 $R/scripts/time_general_rate.py > $O/sq_general.log 2>&1 || echo general counters failed
cat $O/sq_general_1p77/summary.txt
echo collected
