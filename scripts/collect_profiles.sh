#!/bin/bash
# Round profile collection on the GPU box: kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE
# passes for each bench workload.  Output under gpurun_out/prof_<tag>/; summarised into
# profiles/ by scripts/summarise_profiles.py (run in the authoring container).
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for WL in fir decimate resample; do
  O=$R/gpurun_out/prof_${TAG}_${WL}
  mkdir -p $O
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --workload $WL --no-cpu > $O/kt.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu > $O/fetch.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu > $O/write.log 2>&1 || exit 1
  tail -1 $O/kt.log | cut -c1-160
done
# ablation / variant tables from the diagnostic library (regenerable: scripts/ablate.py, scripts/ab_fir.py)
mkdir -p $R/gpurun_out/prof_${TAG}_tables
cd $R
timeout -k 10 300 python3 scripts/ablate.py fir > gpurun_out/prof_${TAG}_tables/fir_variants_ab.txt 2>&1 || exit 1
timeout -k 10 300 python3 scripts/ablate.py resample > gpurun_out/prof_${TAG}_tables/resample_fft_ablation.txt 2>&1 || exit 1
