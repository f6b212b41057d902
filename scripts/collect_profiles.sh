#!/bin/bash
# Round profile collection on the GPU box: kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE
# passes for each bench workload (counters never combined with tracing), SQ/LDS counter passes for
# the two transform kernels, and the variant / ablation tables from the diagnostic library.
# Output under gpurun_out/prof_<tag>_*/; summarised into profiles/ by scripts/summarise_profiles.py
# (run in the authoring container).   usage: scripts/collect_profiles.sh [tag]
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for WL in fir decimate resample; do
  O=$R/gpurun_out/prof_${TAG}_${WL}
  mkdir -p $O
  # the kernel sources these counters belong to (bench.py reports roofline.traffic only while this matches the tree)
  (cd $R && python3 -m simplefe_amd.build --hash $WL) > $O/csrc_hash.txt
  # --no-others: the headline kernel alone (the default line's other_configs legs launch the same
  # kernel name on other shapes, which would mix into the per-name average)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --workload $WL --no-cpu --no-others > $O/kt.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu --no-others > $O/fetch.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu --no-others > $O/write.log 2>&1 || exit 1
  tail -1 $O/kt.log | cut -c1-200
done
# the per-rank launches of the N = 2, 4, 8 channel-sharded job (and the whole job on one GPU), so that
# roofline.traffic is not null in a SCALE line: key fir256_cf32_2p<k>_<c>ch = what ONE rank launches
for SHAPE in "64 30" "32 29" "16 28" "8 27"; do
  set -- $SHAPE
  O=$R/gpurun_out/prof_${TAG}_fir_$1ch
  mkdir -p $O
  (cd $R && python3 -m simplefe_amd.build --hash fir) > $O/csrc_hash.txt
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --channels $1 --log2n $2 --steps 3 --warmup 1 --no-cpu --no-others > $O/fetch.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --channels $1 --log2n $2 --steps 3 --warmup 1 --no-cpu --no-others > $O/write.log 2>&1 || exit 1
done
set --
# the driver's own command shape (all legs in one process)
O=$R/gpurun_out/prof_${TAG}_default
mkdir -p $O
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu > $O/kt.log 2>&1 || exit 1
tail -1 $O/kt.log | cut -c1-200
# SQ / LDS / TCC counters of the two transform kernels
cd $R
bash scripts/prof_pmc.sh prof_${TAG}_sq_fir fir_fft4096 -- python3 $R/bench.py --workload fir --steps 3 --warmup 1 --no-cpu --no-others > /dev/null 2>&1 || exit 1
bash scripts/prof_pmc.sh prof_${TAG}_sq_decimate poly_tiled -- python3 $R/bench.py --workload decimate --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1 || exit 1
bash scripts/prof_pmc.sh prof_${TAG}_sq_resample poly_fft256 -- python3 $R/bench.py --workload resample --steps 3 --warmup 1 --no-cpu > /dev/null 2>&1 || exit 1
# round 4: the general-rate transform-domain kernel (2^28 samples at rate 1.77, 381 taps in 3 phases)
# (round 5: ONE shape per counter file -- rate 1.77 alone, the default dispatch alone; round 4's file averaged 1.77 and 0.77)
export LOG2N=28 GENERAL_ONLY=1 RATES=1.77 DEFAULT_ONLY=1
bash scripts/prof_pmc.sh prof_${TAG}_sq_general poly_gen -- python3 $R/scripts/time_general_rate.py > /dev/null 2>&1 || exit 1
unset LOG2N GENERAL_ONLY RATES DEFAULT_ONLY
# COUNTERS_ONLY=1: stop here -- the kernel-trace / counter passes above are what is stamped with the source hash; the tables
# below are interleaved A/B runs of named variants and stay valid while those variants' code does
if [ -n "$COUNTERS_ONLY" ]; then echo collected counters; exit 0; fi
# ablation / variant tables from the diagnostic library (regenerable: scripts/ablate.py, scripts/ab_fir.py, scripts/ab_rs.py)
mkdir -p $R/gpurun_out/prof_${TAG}_tables
timeout -k 10 300 python3 scripts/ablate.py fir > gpurun_out/prof_${TAG}_tables/fir_variants_ab.txt 2>&1 || exit 1
timeout -k 10 300 python3 scripts/ablate.py resample > gpurun_out/prof_${TAG}_tables/resample_fft_ablation.txt 2>&1 || exit 1
ROUNDS=8 timeout -k 10 300 python3 scripts/ab_fir.py T X W e:8 E e:300 > gpurun_out/prof_${TAG}_tables/fir_walk_vs_tickets.txt 2>&1 || exit 1
# round 3: the three product variants with the straight-line store block (T X W) against the round-2 guarded store loop
# (t x w) and against stores interleaved with the last DFT16 (u y); P = the product's own dispatch (measured choice)
ROUNDS=10 timeout -k 10 300 python3 scripts/ab_fir.py T t X x W w u y P > gpurun_out/prof_${TAG}_tables/fir_store_block.txt 2>&1 || exit 1
ROUNDS=8 timeout -k 10 300 python3 scripts/ab_fir.py e:8 b:8 g:8 G:8 e:300 b:300 g:300 > gpurun_out/prof_${TAG}_tables/fir_pattern_sync.txt 2>&1 || exit 1
timeout -k 10 300 python3 scripts/ab_rs.py s t x l w > gpurun_out/prof_${TAG}_tables/resample_walk_vs_tickets.txt 2>&1 || exit 1
# round 3: what taking work off the LDS could give at most (e no in-wave exchange, z no S0 scatter writes, Z neither, y = z at 3
# workgroups per CU) and the buffer-store epilogue against the round-2 exec-masked one (b)
ROUNDS=10 timeout -k 10 300 python3 scripts/ab_rs.py t b e z Z w y > gpurun_out/prof_${TAG}_tables/resample_lds_bounds.txt 2>&1 || exit 1
timeout -k 10 300 python3 scripts/time_rs_channels.py > gpurun_out/prof_${TAG}_tables/resample_channels.txt 2>&1 || exit 1
LOG2N=28 timeout -k 10 300 python3 scripts/time_general_rate.py > gpurun_out/prof_${TAG}_tables/general_rate.txt 2>&1 || exit 1
timeout -k 10 300 python3 scripts/time_real_fir.py > gpurun_out/prof_${TAG}_tables/fir_real_data.txt 2>&1 || exit 1
timeout -k 10 120 scripts/probes/hbm_mix > gpurun_out/prof_${TAG}_tables/hbm_mix.txt 2>&1
timeout -k 10 200 python3 scripts/time_pipe.py > gpurun_out/prof_${TAG}_tables/host_pipe.txt 2>&1
echo collected
