#!/bin/bash
# Round 5, after the counter summaries (profiles/pmc_r05_*.json) are in the tree: the bench lines (they carry roofline.traffic now), the N > 1
# rehearsals on the one device, the tables that changed this round, and the whole GPU suite once more.  Output: gpurun_out/prof_r05_tables/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r05_tables
mkdir -p $O
cd $R
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --telemetry > $O/bench_driver_shape.json 2> $O/bench_driver_shape.err || echo driver shape failed
cut -c1-250 $O/bench_driver_shape.json
timeout -k 10 300 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || echo default failed
cut -c1-250 $O/bench_default.json
SFE_BENCH_ONE_DEVICE=1 timeout -k 10 600 python3 bench.py --gpus 4 --steps 20 --warmup 5 --no-cpu > $O/bench_4ranks_one_device.json 2> $O/bench_4ranks.err || echo 4 ranks failed
cut -c1-250 $O/bench_4ranks_one_device.json
SFE_BENCH_ONE_DEVICE=1 timeout -k 10 300 python3 bench.py --gpus 8 --single-process --steps 20 --warmup 5 > $O/bench_single_process_8_blocks.json 2> $O/bench_sp8.err || echo single process failed
cut -c1-250 $O/bench_single_process_8_blocks.json
timeout -k 10 300 python3 scripts/ablate.py fir > $O/fir_variants_ab.txt 2>&1 || echo ablate failed
cat $O/fir_variants_ab.txt
timeout -k 10 300 python3 scripts/time_zeros_vs_data.py > $O/zeros_vs_data.txt 2>&1 || echo zeros failed
cat $O/zeros_vs_data.txt
LOG2N=28 timeout -k 10 300 python3 scripts/time_general_rate.py > $O/general_rate.txt 2>&1 || echo general failed
cat $O/general_rate.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
echo collected
