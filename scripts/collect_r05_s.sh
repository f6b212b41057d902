#!/bin/bash
# Round 5: the whole GPU suite on the tree with the register-window interpolator kernel, then the real-stream shapes by the product library
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; echo pytest failed; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 300 python3 scripts/time_real_shapes.py > $O/shapes_real.txt 2>&1 && cat $O/shapes_real.txt
echo collected
