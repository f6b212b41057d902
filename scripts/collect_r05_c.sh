#!/bin/bash
# Round 5, third GPU call: the general-rate kernel before / after the branch-free shares loop in ONE process (two builds of the library,
# interleaved), the full-size general-rate parity tests, the shapes table with the bare mix beside each row, the bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05c
mkdir -p $O
cd $R
ROUNDS=12 timeout -k 10 300 python3 scripts/ab_libs.py simplefe_amd/libsfe_dsp_prev.so simplefe_amd/libsfe_dsp.so general > $O/general_ab.txt 2>&1 || echo ab failed
cat $O/general_ab.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "general_rate" -s > $O/pytest_general.txt 2>&1; echo "pytest rc $?" | tee -a $O/pytest_general.txt
grep -E "general rate|passed|failed" $O/pytest_general.txt
BARE=1 timeout -k 10 600 python3 scripts/time_shapes.py > $O/shapes.txt 2>&1 || echo shapes failed
cat $O/shapes.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || { echo bench failed; tail -5 $O/bench_default.err; }
cut -c1-300 $O/bench_default.json
echo collected
