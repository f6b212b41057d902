#!/bin/bash
# L2 (TCC) or L1/TLB/address-unit (TCP, TA, SQ) counters of the hbm_mix probe kernels, one --pmc pass
# per group of four (kernel-trace only).  Run on a GPU box from the repo root:
#   bash scripts/probes/pmc_hbm_mix.sh gpurun_out/<dir> tcc|tcp
# then scripts/probes/summarise_pmc_hbm_mix.py gpurun_out/<dir>
set -e
out=${1:-gpurun_out/pmc_hbm_mix}
set_=${2:-tcc}
mkdir -p "$out"
hipcc --offload-arch=gfx950 -O3 scripts/probes/hbm_mix.hip -o "$out/hbm_mix"
export HBM_MIX_R=2
if [ "$set_" = tcc ]; then
groups=("TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_TAG_STALL"
        "TCC_SRC_FIFO_FULL TCC_LATENCY_FIFO_FULL TCC_BUBBLE TCC_IB_STALL"
        "TCC_EA0_WRREQ TCC_EA0_WRREQ_64B TCC_NORMAL_WRITEBACK TCC_NORMAL_EVICT"
        "TCC_HIT TCC_MISS TCC_WRITE_REQ TCC_STREAMING_REQ"
        "TCC_WRITE_REQ_LATENCY TCC_EA0_WRREQ_LEVEL TCC_CYCLE TCC_BUSY"
        "TCC_READ_REQ TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL TCC_READ_REQ_LATENCY")
else
groups=("TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_UTCL1_STALL_INFLIGHT_MAX"
        "TCP_UTCL1_STALL_MULTI_MISS TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS TCP_UTCL1_SERIALIZATION_STALL TCP_UTCL1_LFIFO_FULL"
        "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_LFIFO_STALL_CYCLES TCP_RFIFO_STALL_CYCLES"
        "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES"
        "TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_WRITE_REQ TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ"
        "TCP_WRITE_TAGCONFLICT_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_GATE_EN1 TCP_GATE_EN2"
        "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM"
        "SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY")
fi
i=0
for grp in "${groups[@]}"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp -d "$out/p$i" -o p --output-format csv -- "$out/hbm_mix" > "$out/p$i.log" 2>&1 || echo "pass $i failed: $grp"
done
rm -f "$out/hbm_mix"
