#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>_<workload>/ (scripts/collect_profiles.sh) into the committed
summaries: profiles/<tag>/<workload>_kernel_stats.csv, ..._pmc.csv and profiles/pmc_<tag>_<workload>.json
(the file bench.py reads `roofline.traffic` from)."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simplefe_amd.build import csrc_hash  # noqa: E402

tag = next((a for a in sys.argv[1:] if not a.startswith("--")), "r05")
# --tables-only: copy the bench lines and tables of gpurun_out/prof_<tag>_tables/ only -- the counter summaries (profiles/pmc_<tag>_*.json) are left as
# committed (round 5: a second run of this script over the same gpurun_out/ re-stamped them with the hash file of the collection)
TABLES_ONLY = "--tables-only" in sys.argv
KEYS = {"fir": ("fir256_cf32_2p28", "fir_fft4096", 16 * 2 ** 28),
        "decimate": ("decimate8_cf32_2p30", "poly_tiled", 9 * 2 ** 30),
        "resample": ("resample5o3_cf32_2p28", "poly_fft256", None),
        # what one rank of the channel-sharded job launches at N = 1, 2, 4, 8 (bench.py: leg.key)
        "fir_64ch": ("fir256_cf32_2p30_64ch", "fir_fft4096", 16 * 2 ** 30), "fir_32ch": ("fir256_cf32_2p29_32ch", "fir_fft4096", 16 * 2 ** 29),
        "fir_16ch": ("fir256_cf32_2p28_16ch", "fir_fft4096", 16 * 2 ** 28), "fir_8ch": ("fir256_cf32_2p27_8ch", "fir_fft4096", 16 * 2 ** 27)}
os.makedirs(os.path.join(ROOT, "profiles", tag), exist_ok=True)
for wl, (key, ksub, alg) in ({} if TABLES_ONLY else KEYS).items():
    d = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{wl}")
    if not os.path.isdir(d):
        continue
    ks = sorted(glob.glob(d + "/kt/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime)
    if ks:
        shutil.copy(ks[-1], os.path.join(ROOT, "profiles", tag, f"{wl}_kernel_stats.csv"))       # the latest collection
    vals, rows_out = {}, []
    for c in ("fetch", "write"):
        f = sorted(glob.glob(d + f"/{c}/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)
        if not f:
            continue
        for r in csv.DictReader(open(f[-1])):
            rows_out.append({k: r[k] for k in ("Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size",
                                               "VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value",
                                               "Start_Timestamp", "End_Timestamp")})
            if ksub in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    if rows_out:
        with open(os.path.join(ROOT, "profiles", tag, f"{wl}_pmc.csv"), "w") as o:
            w = csv.DictWriter(o, fieldnames=list(rows_out[0].keys()))
            w.writeheader()
            w.writerows(rows_out)
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        fetch = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"])
        write = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"])
        # the hash written on the GPU box beside the counters (collect_profiles.sh); collections without one
        # are stamped with the tree as it is now -- only right if nothing under csrc/ changed since
        hf = os.path.join(d, "csrc_hash.txt")
        kind = "fir" if wl.startswith("fir") else wl          # which kernel's files the stamp covers (simplefe_amd/build.py: KERNEL_FILES)
        stamp = open(hf).read().strip() if os.path.exists(hf) else csrc_hash(kind)
        out = {"workload": key, "round": tag, "kernel_substr": ksub, "csrc_sha256": stamp, "csrc_kind": kind, "FETCH_SIZE_KiB_raw": fetch, "WRITE_SIZE_KiB": write,
               "correction": "gfx950: FETCH_SIZE tallies 128-B requests as 64 B -> x2 (MI355X_MICROARCH.md, HBM); "
                             "WRITE_SIZE exact (calibrated: synth_fill_kernel writes 2 GiB -> 2097152 KiB)",
               "hbm_bytes_per_launch": (2 * fetch + write) * 1024, "algorithmic_bytes_per_launch": alg,
               "how": "separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE) over bench.py --workload <wl> --steps 3 --warmup 1 --no-cpu --no-others"}
        json.dump(out, open(os.path.join(ROOT, "profiles", f"pmc_{tag}_{wl}.json"), "w"), indent=1)
        print(wl, out["hbm_bytes_per_launch"] / 1e9, "GB per launch")

# the driver's command shape, SQ counter summaries and the tables
d = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_default")
ks = sorted(glob.glob(d + "/kt/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime)
if ks:
    shutil.copy(ks[-1], os.path.join(ROOT, "profiles", tag, "default_line_kernel_stats.csv"))
for wl in ("fir", "resample", "decimate", "general"):
    f = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_sq_{wl}", "summary.txt")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(ROOT, "profiles", tag, f"{wl}_sq_counters.txt"))
for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_tables", "*.txt")) + glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_tables", "*.json")):
    shutil.copy(f, os.path.join(ROOT, "profiles", tag, os.path.basename(f)))
