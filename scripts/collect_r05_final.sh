#!/bin/bash
# Round 5, last GPU call on the final tree: everything scripts/collect_r05_lines.sh collects (bench lines with traffic, rehearsals, tables, the whole
# GPU suite) plus the shapes tables (complex with bare mixes, long decimations, real streams).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash scripts/collect_r05_lines.sh
O=$R/gpurun_out/prof_r05_tables
BARE=1 timeout -k 10 600 python3 scripts/time_shapes.py > $O/shapes_main.txt 2>&1 || echo shapes failed
BARE=1 EXTRA=1 SHAPES="by 12,by 13,by 24,by 32,by 48" timeout -k 10 300 python3 scripts/time_shapes.py > $O/shapes_long.txt 2>&1 || echo long failed
timeout -k 10 300 python3 scripts/time_real_shapes.py > $O/shapes_real_final.txt 2>&1 || echo real failed
EXTRA=3 SHAPES="interpolate" timeout -k 10 300 python3 scripts/time_shapes.py > $O/shapes_interp_cplx.txt 2>&1 || echo interp failed
timeout -k 10 300 python3 scripts/time_real_baseline.py > $O/baseline_real.txt 2>&1 || echo real baseline failed
cat $O/baseline_real.txt
timeout -k 10 300 python3 scripts/time_u8_shapes.py > $O/shapes_u8_final.txt 2>&1 || echo u8 failed
REAL=1 timeout -k 10 300 python3 scripts/time_u8_shapes.py >> $O/shapes_u8_final.txt 2>&1 || echo u8 real failed
timeout -k 10 400 python3 scripts/time_real_compiled.py > $O/shapes_real_compiled_final.txt 2>&1 || echo real compiled failed
cut -c1-110 $O/shapes_main.txt; cat $O/shapes_real_final.txt
echo collected final
