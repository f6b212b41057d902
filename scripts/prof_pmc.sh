#!/bin/bash
# Collect PMC passes for one command; each pass its own rocprofv3 run (no tracing combined).
# usage: prof_pmc.sh <outdir-under-gpurun_out> <kernel-substr> -- <program> [args]   (program after --, never a wrapper)
set -e
OUT=$1; KSUB=$2; shift 3
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/$OUT
cd /tmp; export TMPDIR=/tmp
PASSES=(
"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU"
"SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU"
"GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL"
"FETCH_SIZE"
"WRITE_SIZE"
"TCC_HIT_sum TCC_MISS_sum"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $R/gpurun_out/$OUT/p$i -- "$@" > $R/gpurun_out/$OUT/p$i.log 2>&1
done
python3 - "$R/gpurun_out/$OUT" "$KSUB" <<'PY'
import csv,glob,sys,collections
d,ks=sys.argv[1],sys.argv[2]
agg=collections.OrderedDict()
for f in sorted(glob.glob(d+"/p*/**/*_counter_collection.csv",recursive=True)):
    for r in csv.DictReader(open(f)):
        if ks in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
with open(d+"/summary.txt","w") as o:
    for k,v in agg.items():
        line=f"{k:32s} n={len(v):3d} mean={sum(v)/len(v):.6g}"
        print(line); o.write(line+"\n")
PY
