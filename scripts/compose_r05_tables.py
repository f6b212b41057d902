#!/usr/bin/env python3
"""After scripts/collect_r05_final.sh (round 5's last GPU call): put its shapes tables into the files DESIGN.md cites -- profiles/r05/shapes.txt
(main + long decimations + complex interpolators x5 ... x7), the last block of profiles/r05/shapes_real.txt, the header of baseline_real.txt -- then
run scripts/summarise_profiles.py r05 --tables-only and scripts/design_figures.py r05 --write."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = os.path.join(ROOT, "gpurun_out", "prof_r05_tables") + "/"
P = os.path.join(ROOT, "profiles", "r05") + "/"
if not os.path.exists(T + "shapes_main.txt"):
    sys.exit("no fresh collection under gpurun_out/prof_r05_tables/")
hdr = open(P + "shapes.txt").read().split("\n")[:5]
body = lambda f, keep=lambda l: True: [l for l in open(T + f).read().split("\n") if l.strip() and not l.startswith(("#", "shape")) and keep(l)]
main = [l for l in open(T + "shapes_main.txt").read().split("\n") if l.strip()]
open(P + "shapes.txt", "w").write("\n".join(hdr + main + body("shapes_long.txt") + body("shapes_interp_cplx.txt", lambda l: any(k in l for k in ("x5", "x6", "x7")))) + "\n")
s = open(P + "shapes_real.txt").read()
marker = "#### the LAST GPU call"
s = s[:s.index(marker)] + marker + " (final tree -- the window kernel at input steps 1 ... 5 included; the first shape is preceded by 150 warm-up launches)\n" + open(T + "shapes_real_final.txt").read()
open(P + "shapes_real.txt", "w").write(s)
for dst, src, what in (("shapes_u8.txt", "shapes_u8_final.txt", "u8 wire-format input, complex then real streams, the product library's default dispatch"),
                       ("shapes_real_window.txt", "shapes_real_compiled_final.txt", "the ratios with compile-time kernels, real and complex streams, the product library's default dispatch")):
    if os.path.exists(T + src):
        s = open(P + dst).read()
        if marker in s:
            s = s[:s.index(marker)]
        open(P + dst, "w").write(s.rstrip("\n") + "\n" + marker + " (final tree): " + what + "\n" + open(T + src).read())
        os.remove(T + src)
for f in ("shapes_main.txt", "shapes_long.txt", "shapes_interp_cplx.txt", "shapes_real_final.txt"):
    os.remove(T + f)
b = open(T + "baseline_real.txt").read()
if not b.startswith("# scripts/time_real_baseline.py"):
    open(T + "baseline_real.txt", "w").write("# scripts/time_real_baseline.py (round 5, the last GPU call): the three BASELINE workloads on a REAL float32 stream -- libdsp's native type --\n"
                                             "# default dispatch, 200 warm-up launches per row.  (The script's first form, 4 warm-up launches, put the FIR at 0.8980 ms: the chip's post-idle transient.)\n" + b)
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "summarise_profiles.py"), "r05", "--tables-only"])
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "design_figures.py"), "r05", "--write"])
