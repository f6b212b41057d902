#!/bin/bash
# Round 5, eleventh GPU call: the 5/3 resampler's variants and ablations judged by clock and watts, data and zeros
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05k
mkdir -p $O
cd $R
ROUNDS=6 WATTS=1 timeout -k 10 400 python3 scripts/ab_rs.py t e z Z w y > $O/resample_energy_data.txt 2>&1 || echo data failed
cat $O/resample_energy_data.txt
ROUNDS=6 WATTS=1 ZEROS=1 timeout -k 10 300 python3 scripts/ab_rs.py t e Z w > $O/resample_energy_zeros.txt 2>&1 || echo zeros failed
cat $O/resample_energy_zeros.txt
echo collected
