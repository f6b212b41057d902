#!/bin/bash
# Round 5, ninth GPU call: the general-rate kernel with and without its prologue (ablation), and the general-rate tests once more (the kernel's source was touched)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05i
mkdir -p $O
cd $R
timeout -k 10 300 python3 scripts/ab_general.py > $O/general_rate_prologue_ablation.txt 2>&1 || echo ab failed
cat $O/general_rate_prologue_ablation.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_general_rate.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" | tee -a $O/pytest.txt
tail -2 $O/pytest.txt
echo collected
