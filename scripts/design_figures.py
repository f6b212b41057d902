#!/usr/bin/env python3
"""Print the figures DESIGN.md quotes from profiles/<tag>/ in one place, so that a re-collection can be
followed through the prose quickly (tests/test_design_numbers.py then checks the result).
usage: scripts/design_figures.py [tag] [--write]
--write also regenerates DESIGN.md's block of the collection's own figures (between the collection:begin / :end markers)
and the runtime shapes' table of 4.2d (shapes:begin / :end, from profiles/<tag>/shapes.txt):
after a re-collection (scripts/collect_profiles.sh, scripts/summarise_profiles.py, the two bench lines) that is the only
edit the prose needs."""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simplefe_amd.build import fir_kernel_flags  # noqa: E402

tag = next((a for a in sys.argv[1:] if not a.startswith("--")), "r05")
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


# ---- `--write`: regenerate DESIGN.md's block of this collection's figures (between the collection:begin / :end markers)
def _line(name):
    return json.loads(open(os.path.join(P, name + ".json")).read().strip().splitlines()[-1])


def _stats(w):
    r = list(csv.DictReader(open(os.path.join(P, f"{w}_kernel_stats.csv"))))[0]
    return int(r["Calls"]), float(r["AverageNs"])


def _other(j, key):
    return next(o for o in j["other_configs"] if key in o["workload"])


def collection_block():
    drv, dfl = _line("bench_driver_shape"), _line("bench_default")
    rows = []
    keys = {"fir": None, "resample": "381-tap prototype (127", "decimate": "decimate by 8"}
    for w, label in (("fir", "256-tap FIR, 2^28 cf32 (`fir_fft4096_kernel`, register loads)"),
                     ("resample", "resample 5/3, 381 taps, 2^28 cf32 (`poly_fft256_kernel`)"),
                     ("decimate", "decimate ÷8, 64 taps, 2^30 cf32 (`poly_tiled_kernel`)")):
        calls, ns = _stats(w)
        ms = ns / 1e6
        pmc = json.load(open(os.path.join(ROOT, "profiles", f"pmc_{tag}_{w}.json")))
        if keys[w] is None:
            d_ms, d_fr, f_ms, f_fr = drv["roofline"]["kernel_ms"], drv["roofline"]["frac"], dfl["roofline"]["kernel_ms"], dfl["roofline"]["frac"]
        else:
            a, b = _other(drv, keys[w]), _other(dfl, keys[w])
            d_ms, d_fr, f_ms, f_fr = a["ms"], a["frac"], b["ms"], b["frac"]
        sp = lambda v: f"{v:,.0f}".replace(",", " ")
        rows.append(f"| {label} | mean {sp(ns)} ns = {ms:.4f} ms over {calls} launches (`profiles/{tag}/{w}_kernel_stats.csv`) = "
                    f"{ALG[w] / ms / 1e9:.2f} TB/s = **{ALG[w] / ms / 1e9 / 8 * 100:.1f} %** | {d_ms:.4f} ms, `frac` {d_fr:.3f} (`profiles/{tag}/bench_driver_shape.json`) | "
                    f"{f_ms:.4f} ms, `frac` {f_fr:.3f} (`profiles/{tag}/bench_default.json`) | {pmc['hbm_bytes_per_launch'] / 1e9:.2f} GB, "
                    f"+{(pmc['hbm_bytes_per_launch'] / ALG[w] - 1) * 100:.1f} % over algorithmic (`profiles/pmc_{tag}_{w}.json`) |")
    r, c = drv["roofline"], drv["cpu_baseline"]
    o = {k: _other(drv, k) for k in ("381-tap prototype (127", "127-tap", "decimate by 8", "complex-tap", "general-rate")}
    o["381-tap"] = o["381-tap prototype (127"]
    ch64 = [x for x in drv["other_configs"] if "64 channel" in x["workload"]][0]
    head = (f"The driver's command shape measured by the builder (`python bench.py --gpus 1 --steps 20 --warmup 5`, `profiles/{tag}/bench_driver_shape.json`): "
            f"**{drv['value']:,.0f} MS/s, `ms_per_step` {drv['ms_per_step']:.3f}, `roofline.frac` {r['frac']:.3f}** (kernel {r['kernel_ms']:.4f} ms mean, "
            f"{r['kernel_ms_min']:.4f} min, {r['kernel_ms_max']:.4f} max; the three plain pairs timed before: {', '.join('%.4f' % v for v in drv['config'].get('pairs', {}).get('pairs_timed_ms', []))} ms, the median one kept; `traffic` {r['traffic'] / 1e9:.3f} GB; variant: {r['variant']['ran']}, {r['variant'].get('chosen_by', '')}), parity "
            f"{drv['parity']['rel_rms_max']:.1e}; `other_configs`: resample 5/3 {o['381-tap']['ms']:.4f} ms ({o['381-tap']['frac']:.3f}), with the 127-tap prototype "
            f"{o['127-tap']['ms']:.4f} ms ({o['127-tap']['frac']:.3f}), decimate ÷8 {o['decimate by 8']['ms']:.4f} ms ({o['decimate by 8']['frac']:.3f}; plain allocations: whichever placement mode the process drew), "
            f"general rate 1.77 {o['general-rate']['ms']:.4f} ms ({o['general-rate']['frac']:.3f}), 64 channels × 2^24 "
            f"{ch64['ms']:.2f} ms ({ch64['frac']:.3f}), complex taps {o['complex-tap']['ms']:.4f} ms ({o['complex-tap']['frac']:.3f}), every parity check green; "
            f"`cpu_baseline` {c['value']:.0f} MS/s on one thread, {c['all_cores']['value']:.0f} MS/s on the box's {c['host_cores']} host cores"
            + (f", BASELINE configs[0] (CPU only) {c['configs0_cpu_only']['value']:.0f} real MS/s" if "configs0_cpu_only" in c else "") + ".")
    head = head.replace(f"{drv['value']:,.0f}", f"{drv['value']:,.0f}".replace(",", " ")).replace(f"{r['traffic'] / 1e9:.3f} GB", f"{r['traffic'] / 1e9:.2f} GB")
    earlier = []
    for f in (sorted(os.listdir(os.path.join(P, "earlier"))) if os.path.isdir(os.path.join(P, "earlier")) else []):
        if f.endswith("bench_driver_shape.json"):
            j = json.loads(open(os.path.join(P, "earlier", f)).read().strip().splitlines()[-1])
            earlier.append(f"`{f.split('_')[0]}` {j['value']:,.0f} MS/s / {j['roofline']['frac']:.3f}".replace(",", " "))
    tail = ("Other collections of the round, other boxes of the pool, same product kernels (`c6` was made AFTER the one above, on a slower box) (`profiles/" + tag + "/earlier/`, driver shape, MS/s / `frac`): "
            + ", ".join(earlier) + ".") if earlier else ""
    table = ("| kernel | under rocprofv3 (`bench.py --workload … --no-others --no-cpu`) | un-profiled, driver-shape line | un-profiled, default line | PMC traffic |\n"
             "|---|---|---|---|---|\n" + "\n".join(rows))
    return head + "\n\n" + table + "\n\n" + tail


def shapes_block():
    """DESIGN.md 4.2d's table (between the shapes:begin / :end markers) from profiles/<tag>/shapes.txt (BARE=1 scripts/time_shapes.py)"""
    rows = ["| shape | U | step | ms (default dispatch) | `frac` | ms, exact mode | the shape's bare read : write mix |", "|---|---|---|---|---|---|---|"]
    for line in open(os.path.join(P, "shapes.txt")):
        m = re.match(r"(.{28}) +(\d+) +(\d+) +[\d.]+ +([\d.]+) +[\d.]+ +([\d.]+) +([\d.]+) ", line)
        if not m or line.startswith(("#", "shape")):
            continue
        bare = re.search(r"bare ([\d.]+) / ([\d.]+) / (\d+) %", line)
        mix = f"{bare.group(1)} ms, `frac` {bare.group(2)}: the kernel at {bare.group(3)} % of it" if bare else "n/a (the probe's mixes end at 8 : 1)"
        rows.append(f"| {m.group(1).strip()} | {m.group(2)} | {m.group(3)} | {m.group(4)} | {m.group(5)} | {m.group(6)} | {mix} |")
    return "\n" + "\n".join(rows)


if "--write" in sys.argv:
    path = os.path.join(ROOT, "DESIGN.md")
    text = open(path).read()
    for (b, e), make in ((("<!-- collection:begin -->", "<!-- collection:end -->"), collection_block),
                         (("<!-- shapes:begin -->", "<!-- shapes:end -->"), shapes_block)):
        i, j = text.index(b) + len(b), text.index(e)
        block = make()                          # may raise: nothing has been opened for writing yet
        text = text[:i] + "\n" + block + "\n" + text[j:]
    open(path, "w").write(text)
    print("DESIGN.md: collection and shapes blocks rewritten")
