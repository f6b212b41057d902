#!/bin/bash
# How the three bulk kernels' stores reach HBM: write requests by size (TCC_EA_WRREQ, _64B) and read requests by size,
# one rocprofv3 --pmc pass per workload (counters only, no tracing).  usage (on the GPU box): scripts/probes/pmc_writes.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for WL in fir resample decimate; do
  O=$R/gpurun_out/pmc_writes_$WL
  mkdir -p $O
  timeout -k 10 200 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/p -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu --no-others > $O/log 2>&1
  timeout -k 10 200 rocprofv3 --pmc TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_LEVEL_sum --output-format csv -d $O/p/s -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu --no-others >> $O/log 2>&1
  python3 - $O $WL <<'PY'
import csv,glob,sys,collections
d,wl=sys.argv[1],sys.argv[2]
ks={"fir":"fir_fft4096","resample":"poly_fft256","decimate":"poly_tiled"}[wl]
agg=collections.OrderedDict()
for f in glob.glob(d+"/p/**/*_counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if ks in r["Kernel_Name"]:
            agg.setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
m={k:sum(v)/len(v) for k,v in agg.items()}
print(wl, " ".join(f"{k}={v:.4g}" for k,v in m.items()))
if m.get("TCC_EA0_WRREQ_sum"):
    print(f"   writes: {100*m.get('TCC_EA0_WRREQ_64B_sum',0)/m['TCC_EA0_WRREQ_sum']:.1f} % of write requests are 64-byte; reads: {100*m.get('TCC_EA0_RDREQ_32B_sum',0)/max(m.get('TCC_EA0_RDREQ_sum',1),1):.1f} % of read requests are 32-byte")
PY
done
