#!/bin/bash
# Round 5: the LDS-DMA runtime kernel generalised to real float32 streams and wide reads: the parity tests, the real-stream shapes before / after, the complex shapes once more
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05o
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_dropin.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" | tee -a $O/pytest.txt
tail -4 $O/pytest.txt
timeout -k 10 300 python3 scripts/time_real_shapes.py > $O/shapes_real.txt 2>&1 || echo real failed
cat $O/shapes_real.txt
BARE=1 SHAPES="by 6,by 7,by 16,7/4,7/3,4/5" timeout -k 10 300 python3 scripts/time_shapes.py > $O/shapes_cplx.txt 2>&1 || echo cplx failed
cut -c1-100 $O/shapes_cplx.txt
echo collected
