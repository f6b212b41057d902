#!/bin/bash
# counters of the decimator kernel in fresh processes, to compare the processes that land in
# the fast mode with those that land in the slow one.  Counters only (no tracing), one pass per rocprofv3 run,
# the program directly after `--`.   usage: scripts/dec_modes_pmc.sh [processes-per-pass]
NP=${1:-5}
ONLY=${2:-}          # e.g. 2: that pass alone
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/dec_modes_pmc
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
PASSES=(
"TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE"
"TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum"
"TCC_EA0_RDREQ TCC_EA0_WRREQ"
)
p=0
for P in "${PASSES[@]}"; do
  p=$((p+1))
  if [ -n "$ONLY" ] && [ "$ONLY" != "$p" ]; then continue; fi
  for i in $(seq 1 $NP); do
    timeout -k 10 200 rocprofv3 --pmc $P --output-format csv json -d $O/p${p}_$i -- python3 $R/scripts/dec_modes.py one > $O/p${p}_$i.log 2>&1 || { echo "pass $p run $i failed"; tail -3 $O/p${p}_$i.log; }
    tail -1 $O/p${p}_$i.log | cut -c1-220
  done
done
python3 $R/scripts/dec_modes_pmc_summary.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
