#!/usr/bin/env python3
"""Table of the TCC counters scripts/probes/pmc_hbm_mix.sh collected: one row per probe kernel and
grid, counters averaged over its dispatches, duration from the same dispatches' timestamps."""
import csv
import glob
import re
import sys
from collections import defaultdict

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_hbm_mix"
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(d + "/p*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_" not in k:
            continue
        m = re.search(r"(k_\w+)<(.*)>", k)
        name = (m.group(1) + "<" + m.group(2).replace("__attribute__((ext_vector_type(", "v").replace(")))", "").replace("float ", "f") + ">") if m else k
        key = (name, int(r["Grid_Size"]) // 256)
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[key]["us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
cols = sorted({c for v in acc.values() for c in v} - {"us"})
print("kernel\tgrid\tus\t" + "\t".join(cols))
for key in sorted(acc):
    v = acc[key]
    print("%s\t%d\t%.1f\t" % (key[0], key[1], sum(v["us"]) / len(v["us"])) + "\t".join("%.4g" % (sum(v[c]) / len(v[c])) if c in v else "" for c in cols))
