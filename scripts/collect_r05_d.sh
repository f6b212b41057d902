#!/bin/bash
# Round 5, fourth GPU call: the whole GPU suite after the FIR kernel lost its five legacy template switches, the bench line with
# every input verified up-front, and the product's three variants beside the bare patterns with clocks and watts.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05d
mkdir -p $O
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --telemetry > $O/bench_default.json 2> $O/bench_default.err || { echo bench failed; tail -5 $O/bench_default.err; }
cut -c1-300 $O/bench_default.json
ROUNDS=6 WATTS=1 timeout -k 10 300 python3 scripts/ab_fir.py T X W E e:8 e:300 > $O/fir_variants_watts.txt 2>&1 || echo ab_fir failed
cat $O/fir_variants_watts.txt
echo collected
