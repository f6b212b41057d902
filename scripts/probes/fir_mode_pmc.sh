#!/bin/bash
# Counters of the FIR's launches on a fast-type and a slow-type input in ONE process (scripts/probes/fir_mode_py.py pmc):
# each pass its own rocprofv3 run, several processes per pass (a fast first input turns up in ~40 % of fresh processes).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/fir_mode_pmc
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
PASSES=(
"GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum"
"GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum"
"GRBM_GUI_ACTIVE TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  for k in 1 2 3 4; do
    timeout -k 10 120 rocprofv3 --pmc $P --output-format csv -d $O/p${i}_$k -- python3 $R/scripts/probes/fir_mode_py.py pmc > $O/p${i}_$k.log 2>&1 || exit 1
  done
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
names = ["in1->out1", "in2->out2", "in1->out2", "in2->out1"]
with open(d + "/summary.txt", "w") as o:
    for run in sorted(glob.glob(d + "/p*_?")):
        log = [l for l in open(run + ".log") if l.startswith("pmc order")]
        by = collections.OrderedDict()
        for f in sorted(glob.glob(run + "/**/*_counter_collection.csv", recursive=True)):
            for r in csv.DictReader(open(f)):
                if "fir_fft4096" in r["Kernel_Name"]:
                    by.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        o.write("%s  %s" % (run.split("/")[-1], log[0] if log else "(no timing line)\n"))
        for c, v in by.items():
            v = [x for _, x in sorted(v)][-16:]
            o.write("    %-40s %s\n" % (c, "  ".join("%s %.5g" % (names[q], sum(v[4 * q:4 * q + 4]) / 4) for q in range(4))))
print(open(d + "/summary.txt").read())
PY
