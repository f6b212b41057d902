#!/usr/bin/env python3
"""Summarise scripts/dec_modes_pmc.sh: per profiled process, the decimator kernel's median duration under the
profiler and its counters per launch (csv); for the raw per-instance pass, the spread over the L2 channels (json)."""
import collections
import csv
import glob
import json
import os
import sys

import numpy as np

d = sys.argv[1]
for run in sorted(glob.glob(os.path.join(d, "p*_*"))):
    if not os.path.isdir(run):
        continue
    vals, dur = collections.OrderedDict(), {}
    for f in glob.glob(run + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "poly_tiled" not in r["Kernel_Name"]:
                continue
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    log = open(run + ".log").read().strip().splitlines()
    own = log[-1].split("median")[1].split()[0] if log and "median" in log[-1] else "?"
    ds = np.array(sorted(dur.values())) if dur else np.array([0.0])
    print(f"{os.path.basename(run):8s} kernel under the profiler: median {np.median(ds):.4f} ms (n={len(ds)}); the script's own events: {own} ms")
    for k, v in vals.items():
        print(f"    {k:48s} {np.mean(v):.6g}")
    # per-instance values of the raw counters
    for f in glob.glob(run + "/**/*_results.json", recursive=True):
        try:
            j = json.load(open(f))["rocprofiler-sdk-tool"][0]
        except Exception as e:      # noqa: BLE001
            print("    (json not readable: %s)" % e)
            continue
        names = {}
        for c in j.get("counters", []):
            names[c["id"]["handle"]] = c["name"]
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        kern = {k["kernel_id"]: k.get("formatted_kernel_name", k.get("kernel_name", "")) for k in j.get("kernel_symbols", [])}
        for rec in j.get("callback_records", {}).get("counter_collection", []):
            kid = rec["dispatch_data"]["dispatch_info"]["kernel_id"]
            if "poly_tiled" not in kern.get(kid, ""):
                continue
            for r in rec["records"]:
                cid = r["counter_id"]["handle"]
                per[names.get(cid, str(cid))][rec["dispatch_data"]["dispatch_info"]["dispatch_id"]].append(r["value"])
        for name, disp in per.items():
            arr = np.array([v for v in disp.values() if len(v) == len(next(iter(disp.values())))])
            if arr.ndim != 2 or arr.shape[1] < 2:
                continue
            m = arr.mean(axis=0)
            print(f"    {name}: {arr.shape[1]} instances, per launch: sum {m.sum():.4g}  min {m.min():.4g}  max {m.max():.4g}  "
                  f"max/mean {m.max() / m.mean():.3f}  std/mean {m.std() / m.mean():.4f}")
