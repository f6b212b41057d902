#!/usr/bin/env python3
"""Print the figures DESIGN.md quotes from profiles/<tag>/ in one place, so that a re-collection can be
followed through the prose quickly (tests/test_design_numbers.py then checks the result).
usage: scripts/design_figures.py [tag]"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simplefe_amd.build import fir_kernel_flags  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
P = os.path.join(ROOT, "profiles", tag)
ALG = {"fir": 16 * 2 ** 28, "decimate": 9 * 2 ** 30, "resample": 8 * 2 ** 28 + 8 * 161061273}

print("== kernel stats (rocprofv3 --kernel-trace --stats)")
for w in ("fir", "resample", "decimate"):
    f = os.path.join(P, f"{w}_kernel_stats.csv")
    if not os.path.exists(f):
        continue
    for r in list(csv.DictReader(open(f)))[:3]:
        fl = fir_kernel_flags(r["Name"])
        what = " ".join(k for k, v in fl.items() if v is True) if fl else re.sub(r".*::", "", r["Name"])[:40]
        ms = float(r["AverageNs"]) / 1e6
        print(f"  {w:9s} {what:45s} calls {r['Calls']:>4s}  mean {float(r['AverageNs']):.0f} ns = {ms:.4f} ms"
              f"  -> {ALG[w] / ms / 1e9:.3f} TB/s = {ALG[w] / ms / 1e9 / 8 * 100:.1f} %  (min {float(r['MinNs']) / 1e6:.4f} max {float(r['MaxNs']) / 1e6:.4f})")
print("== PMC traffic")
for w in ("fir", "resample", "decimate"):
    f = os.path.join(ROOT, "profiles", f"pmc_{tag}_{w}.json")
    if os.path.exists(f):
        j = json.load(open(f))
        print(f"  {w:9s} {j['hbm_bytes_per_launch'] / 1e9:.3f} GB = {j['hbm_bytes_per_launch'] / ALG[w]:.3f} x algorithmic   stamp {j.get('csrc_sha256', '')[:12]}")
print("== bench lines")
for name in ("bench_driver_shape", "bench_default", "bench_2ranks_one_device"):
    f = os.path.join(P, name + ".json")
    if not os.path.exists(f):
        continue
    j = json.loads(open(f).read().strip().splitlines()[-1])
    r = j["roofline"]
    print(f"  {name}: {j['value']:.0f} MS/s  ms_per_step {j['ms_per_step']:.4f}  frac {r['frac']:.4f}  kernel {r['kernel_ms']:.4f} "
          f"(min {r.get('kernel_ms_min', 0):.4f} max {r.get('kernel_ms_max', 0):.4f} std {r.get('kernel_ms_std', 0):.4f})  traffic {r.get('traffic')}  "
          f"variant {r.get('variant', {}).get('ran')}  scaling {j['scaling']}")
    for o in j.get("other_configs", []):
        if "workload" in o:
            print(f"      {o['workload'][:72]:72s} {o['ms']:.4f} ms  frac {o['frac']:.3f}  {o.get('variant', '')}")
    if "cpu_baseline" in j:
        c = j["cpu_baseline"]
        print(f"      cpu_baseline {c['value']:.1f} MS/s on 1 thread, {c['all_cores']['value']:.1f} on {c['host_cores']} cores")
print("== tables")
for f in sorted(os.listdir(P)):
    if f.endswith(".txt") and f != "hbm_mix.txt" and "sq_counters" not in f:
        print(f"-- {f}")
        for line in open(os.path.join(P, f)):
            if re.search(r"\d\.\d{3,4} ms|ms per call|MS/s", line):
                print("   " + line.rstrip()[:150])
print("== counters per wave per transform / pass")
for w, units in (("fir", 69906 * 4), ("resample", 116207 * 4)):
    f = os.path.join(P, f"{w}_sq_counters.txt")
    if os.path.exists(f):
        vals = dict((m.group(1), float(m.group(2))) for m in (re.match(r"(\S+)\s+n=\s*\d+ mean=(\S+)", l) for l in open(f)) if m)
        print(f"  {w}: " + "  ".join(f"{k[9:]} {vals[k] / units:.0f}" for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM") if k in vals)
              + f"   WAIT_INST_LDS {vals.get('SQ_WAIT_INST_LDS', 0):.3g}  BANK_CONFLICT {vals.get('SQ_LDS_BANK_CONFLICT', 0):.0f}")
