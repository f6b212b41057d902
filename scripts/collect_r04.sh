#!/bin/bash
# Round 4's bench lines and tables (run after COUNTERS_ONLY=1 scripts/collect_profiles.sh r04, same tree).
# Output under gpurun_out/prof_r04_tables/; copied into profiles/r04/ in the authoring container.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r04_tables
mkdir -p $O
cd $R
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/bench_driver_shape.err || exit 1
timeout -k 10 400 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
SFE_BENCH_ONE_DEVICE=1 timeout -k 10 500 python3 bench.py --gpus 4 --steps 20 --warmup 5 --no-cpu > $O/bench_4ranks_one_device.json 2> $O/bench_4ranks_one_device.err || exit 1
SFE_BENCH_ONE_DEVICE=1 timeout -k 10 300 python3 bench.py --gpus 8 --single-process --steps 20 --warmup 5 > $O/bench_single_process_8_blocks.json 2> $O/bench_single_process.err || exit 1
timeout -k 10 300 python3 bench.py --gpus 1 --single-process --steps 20 --warmup 5 > $O/bench_single_process_1_block.json 2>> $O/bench_single_process.err || exit 1
timeout -k 10 300 python3 scripts/time_shapes.py > $O/shapes.txt 2>&1 || exit 1
timeout -k 10 120 scripts/probes/hbm_mix i > $O/hbm_mix_interpolators.txt 2>&1
LOG2N=28 timeout -k 10 300 python3 scripts/time_general_rate.py > $O/general_rate.txt 2>&1 || exit 1
timeout -k 10 200 python3 scripts/time_real_general.py > $O/general_rate_real.txt 2>&1
timeout -k 10 200 python3 scripts/time_block_api.py > $O/block_api_latency.txt 2>&1
timeout -k 10 200 python3 scripts/time_pipe.py > $O/host_pipe.txt 2>&1
timeout -k 10 200 python3 scripts/time_zeros_vs_data.py > $O/zeros_vs_data.txt 2>&1
echo collected
