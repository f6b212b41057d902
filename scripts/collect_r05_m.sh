#!/bin/bash
# Round 5, fifteenth GPU call: the whole GPU suite with poly_rt_dma.hip in the product, the shapes table regenerated (bare mixes beside it), the driver's line
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05m
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" | tee -a $O/pytest_gpu.txt
tail -3 $O/pytest_gpu.txt
BARE=1 timeout -k 10 600 python3 scripts/time_shapes.py > $O/shapes.txt 2>&1 || echo shapes failed
cut -c1-200 $O/shapes.txt
BARE=1 EXTRA=1 SHAPES="by 12,by 13,by 24,by 32,by 48" timeout -k 10 300 python3 scripts/time_shapes.py > $O/shapes_long.txt 2>&1 || echo long failed
cut -c1-200 $O/shapes_long.txt
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_shape.json 2> $O/bench.err || echo bench failed
cut -c1-200 $O/bench_driver_shape.json
echo collected
