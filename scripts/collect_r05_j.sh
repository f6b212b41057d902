#!/bin/bash
# Round 5, tenth GPU call: what ONE rank of the channel-sharded job launches at N = 1, 2, 4, 8 (64, 32, 16, 8 channels x 2^24), timed on the one GPU:
# how much of the per-GPU rate is left when the launch shrinks to 0.4 ms?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05j
mkdir -p $O
cd $R
for SHAPE in "64 30" "32 29" "16 28" "8 27" "4 26"; do
  set -- $SHAPE
  timeout -k 10 300 python3 bench.py --channels $1 --log2n $2 --steps 50 --warmup 10 --no-cpu --no-others > $O/bench_$1ch.json 2> $O/bench_$1ch.err || echo failed $1
  python3 - $O/bench_$1ch.json <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j['roofline']
print(j['config']['channels_per_gpu'], 'channels:', 'kernel_ms', round(r['kernel_ms'],4), 'min', round(r['kernel_ms_min'],4), 'frac', round(r['frac'],4), 'ms_per_step', round(j['ms_per_step'],4), 'variant', r['variant']['ran'], r['variant'].get('median_ms'))
PY
done
echo collected
