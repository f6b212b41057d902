#!/usr/bin/env python3
"""Where and when a launch of the FIR access-pattern kernel did its work: per-workgroup start/end
times, transforms done, XCD and CU (diagnostic library, SFE_FIR_TRACE).  Usage on a GPU box:
  python scripts/probes/fir_trace.py e:8 E e:300
Prints, per variant: launch span, and per XCD the transforms done, the first start and the last end
(µs from the launch's first start) -- a static share that finishes late on one XCD is a tail."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simplefe_amd import build, lib  # noqa: E402
lib.LIB_PATH = build.build_lib(diag=True)
from simplefe_amd import api, synth  # noqa: E402

n = 1 << int(os.environ.get("LOG2N", "28"))
x = api.DeviceArray(2 * n)
x.fill_synth(synth.SEED)
y = api.DeviceArray(2 * n)
f = api.Fir(synth.taps_cfg2(), data_complex=True, algo=lib.FIR_ALGO_FFT)
path = "/tmp/fir_trace.bin"
for v in sys.argv[1:] or ["e:8", "E", "e:300"]:
    vq = v.split("%")                       # e:8%7 = loads/stores gated on bit 7 of the device clock (SFE_FIR_GATE)
    os.environ.pop("SFE_FIR_GATE", None)
    if len(vq) > 1:
        os.environ["SFE_FIR_GATE"] = vq[1]
    vo = vq[0].split("@")                       # e:300@2 = scattered order (SFE_FIR_ORDER)
    os.environ.pop("SFE_FIR_ORDER", None)
    if len(vo) > 1:
        os.environ["SFE_FIR_ORDER"] = vo[1]
    vv = vo[0].split(":")
    os.environ["SFE_FIR_VARIANT"] = vv[0]
    os.environ["SFE_FIR_TGROUPS"] = "8"
    if len(vv) > 1:
        os.environ["SFE_FIR_WG_PER_CU"] = vv[1]
    else:
        os.environ.pop("SFE_FIR_WG_PER_CU", None)
    os.environ.pop("SFE_FIR_TRACE", None)
    for _ in range(5):
        f.process_stream(x, y, n)
    os.environ["SFE_FIR_TRACE"] = path
    f.process_stream(x, y, n)
    os.environ.pop("SFE_FIR_TRACE", None)
    tr = np.fromfile(path, dtype=np.uint64).reshape(-1, 5)
    tr = tr[tr[:, 1] > 0]
    t0 = tr[:, 0].min()
    st = (tr[:, 0] - t0) / 100.0
    en = (tr[:, 1] - t0) / 100.0
    cnt = tr[:, 2].astype(np.int64)
    xcc = tr[:, 3].astype(np.int64)
    hw = tr[:, 4].astype(np.int64)
    cu = ((hw >> 8) & 0xF) + 16 * ((hw >> 12) & 1) + 32 * ((hw >> 13) & 7)      # cu_id, sh_id, se_id
    print(f"== {v}: {len(tr)} workgroups, {cnt.sum()} transforms, span {en.max():.1f} us; work-groups with work: {np.count_nonzero(cnt)}")
    for g in sorted(set(xcc)):
        m = xcc == g
        w = m & (cnt > 0)
        print(f"   XCD {g}: {cnt[m].sum():6d} transforms  first start {st[m].min():7.1f}  last start {st[w].max():7.1f}  "
              f"first end {en[w].min():7.1f}  last end {en[w].max():7.1f}  distinct CUs {len(set(cu[m]))}")
    # rate over time from the per-transform stamps (rows landed, stores issued)
    td = np.fromfile(path + ".done", dtype=np.uint64)
    tc = np.fromfile(path + ".clk", dtype=np.uint64)
    # shader clock (s_memtime, one counter per XCD) against the 100 MHz clock, per 100 us, on XCD 0:
    # in every variant here transform i runs on XCD i % 8
    sel = np.arange(len(td)) % 8 == 0
    ok = sel & (td >= t0)
    o = np.argsort(td[ok])
    wall, clk = td[ok][o].astype(np.float64), tc[ok][o].astype(np.float64)
    e100 = np.arange(wall[0], wall[-1], 10000.0)
    idx = np.searchsorted(wall, e100)
    idx = idx[idx < len(wall)]
    mhz = np.diff(clk[idx]) / np.diff(wall[idx]) * 100.0
    print("   shader clock per 100 us (MHz, s_memtime against s_memrealtime, XCD 0): " + " ".join("%d" % m for m in mhz))
    td = (td[td >= t0] - t0) / 100.0          # stamps older than this launch belong to transforms it skipped (edges)
    h, _ = np.histogram(td, bins=np.arange(0, td.max() + 25, 25.0))
    print("   transforms per 25 us: " + " ".join("%d" % r for r in h))
    life = (en - st)[cnt > 0] / cnt[cnt > 0]
    print("   transforms per workgroup: " + " ".join("%d:%d" % (a_, b_) for a_, b_ in zip(*np.unique(np.minimum(cnt, 100) // 10 * 10, return_counts=True))))
    print("   us per transform per workgroup: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" %
          (life.min(), np.percentile(life, 10), np.median(life), np.percentile(life, 90), life.max()))
