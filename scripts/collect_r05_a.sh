#!/bin/bash
# Round 5, first GPU call: the test suite, the driver's bench line with --telemetry, the FIR energy table (ms + rocm-smi clock and watts +
# the in-kernel clock per variant, on data and on zeros), the power probe for resample / decimate, and the general-rate kernel's counters
# for ONE shape.  Output under gpurun_out/r05a/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" | tee -a $O/pytest.txt
tail -3 $O/pytest.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --telemetry > $O/bench_default.json 2> $O/bench_default.err || { echo bench failed; tail -5 $O/bench_default.err; }
cut -c1-400 $O/bench_default.json
ROUNDS=6 WATTS=1 timeout -k 10 400 python3 scripts/ab_fir.py T X W T:4 T:3 X:4 X:3 L M N l n e:8 > $O/fir_energy_data.txt 2>&1 || echo ab_fir data failed
cat $O/fir_energy_data.txt
ROUNDS=6 WATTS=1 ZEROS=1 timeout -k 10 300 python3 scripts/ab_fir.py T X L N e:8 > $O/fir_energy_zeros.txt 2>&1 || echo ab_fir zeros failed
cat $O/fir_energy_zeros.txt
timeout -k 10 200 python3 scripts/probes/fir_power.py resample > $O/power_resample.txt 2>&1 || echo power resample failed
timeout -k 10 200 python3 scripts/probes/fir_power.py decimate > $O/power_decimate.txt 2>&1 || echo power decimate failed
cat $O/power_resample.txt $O/power_decimate.txt
export LOG2N=28 GENERAL_ONLY=1 RATES=1.77 DEFAULT_ONLY=1
bash scripts/prof_pmc.sh r05a/sq_general_1p77 poly_gen -- python3 $R/scripts/time_general_rate.py > $O/sq_general.log 2>&1 || echo general counters failed
cat $O/sq_general_1p77/summary.txt
echo collected
