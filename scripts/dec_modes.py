#!/usr/bin/env python3
"""the decimator (decimate by 8, 64 taps, 2^30 cf32 -> 2^27, poly_tiled_kernel<8,1>) runs in
two modes ~7 % apart that flip from process to process on one box.  What decides the mode?

    dec_modes.py one              one fresh process: its buffers' addresses and the kernel's median time
    dec_modes.py many [N]         N fresh processes (default 20), one line each, then the histogram
    dec_modes.py sweep            ONE process, one arena: input / output bases moved by k x 4 KiB, k x 32 KiB,
                                  k x 2 MiB (the reference point re-timed between groups)
    dec_modes.py realloc          ONE process: free and re-allocate the two buffers, with other allocations
                                  in between so that they land elsewhere
    dec_modes.py windows          ONE process, DIAGNOSTIC library: the resident workgroups reading 1, 2, 4, ... separate
                                  windows of the stream (interleaved A/B) -- does spreading the window remove the slow mode?
Product library, product kernel (windows: the diagnostic flavour of the same kernel); HIP events on the launch
stream; 3 launches per sample.  `one` also samples the memory / fabric / shader clock levels while launches are in flight."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from simplefe_amd import lib  # noqa: E402
if len(sys.argv) > 1 and sys.argv[1] in ("windows", "tlb", "tlbmany"):
    from simplefe_amd import build
    lib.LIB_PATH = build.build_lib(diag=True)
from simplefe_amd import api, synth  # noqa: E402

N = 1 << 30
CAP = N // 8 + 8
ALG = 9.0 * N


def frac(ms):
    return ALG / (ms * 1e-3) / 8e12


def time_at(r, t, pin, pout, rounds=9, warm=6):
    for _ in range(warm):
        r.process_stream(pin, N, pout, CAP, 8.0)
    v = []
    for _ in range(rounds):
        t.start()
        for _ in range(3):
            r.process_stream(pin, N, pout, CAP, 8.0)
        t.stop()
        v.append(t.elapsed_ms() / 3)
    return float(np.median(v)), float(np.min(v)), float(np.max(v))


def handle():
    return api.Rs(synth.taps_cfg4(), 1, 4096, mode=lib.RS_DECIMATE, data_complex=True), api.Timer()


def clocks():
    """current DPM level of the memory, fabric, SoC and shader clocks (sysfs), sampled while the GPU is busy"""
    import glob
    out = []
    for name in ("pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "pp_dpm_sclk"):
        for f in sorted(glob.glob("/sys/class/drm/card*/device/" + name))[:1]:
            try:
                cur = [l.strip() for l in open(f).read().splitlines() if "*" in l]
                out.append(name[7:] + " " + (cur[0].replace(" ", "") if cur else "?"))
            except OSError:
                out.append(name[7:] + " n/a")
    return ", ".join(out) if out else "clocks n/a"


def one():
    x = api.DeviceArray(2 * N)
    x.fill_synth(synth.SEED)
    y = api.DeviceArray(2 * CAP)
    r, t = handle()
    # burn ~150 ms first: the chip's start-up transient is not the question here
    med, lo, hi = time_at(r, t, x.ptr, y.ptr, rounds=15, warm=60)
    for _ in range(300):                      # ~0.45 s of launches in flight while the clock levels are read
        r.process_stream(x.ptr, N, y.ptr, CAP, 8.0)
    clk = clocks()
    api.sync()
    print(f"in {x.ptr:#014x} out {y.ptr:#014x}  in%2M {x.ptr % (2 << 20):#x} out%2M {y.ptr % (2 << 20):#x} "
          f"(out-in)%1G {(y.ptr - x.ptr) % (1 << 30):#x}  median {med:.4f} ms  min {lo:.4f}  max {hi:.4f}  frac {frac(med):.3f}  [{clk}]", flush=True)


def many(n):
    meds = []
    for i in range(n):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], capture_output=True, text=True, timeout=300)
        line = (out.stdout.strip().splitlines() or ["(no output) " + out.stderr[-200:]])[-1]
        print(f"process {i:2d}: {line}", flush=True)
        if "median" in line:
            meds.append(float(line.split("median")[1].split()[0]))
    a = np.array(meds)
    print(f"# {len(a)} processes: min {a.min():.4f}  median {np.median(a):.4f}  max {a.max():.4f} ms; "
          f"<= 1.50 ms: {(a <= 1.50).sum()}, 1.50-1.55: {((a > 1.50) & (a <= 1.55)).sum()}, > 1.55: {(a > 1.55).sum()}")


def sweep():
    slack = 64 << 20
    arena = api.DeviceArray((8 * N + 8 * CAP + 3 * slack) // 4)
    base_in, base_out = arena.ptr, arena.ptr + 8 * N + slack
    L = lib.load()
    api.check(L.sfe_dsp_synth_fill(base_in, 2 * N + (slack // 4), synth.SEED, 0, 0, None))
    r, t = handle()
    print(f"# arena {arena.ptr:#x}; reference: input at +0, output at +8 GiB + 64 MiB")
    ref = time_at(r, t, base_in, base_out, warm=60)
    print(f"reference                       median {ref[0]:.4f}  min {ref[1]:.4f}  max {ref[2]:.4f}  frac {frac(ref[0]):.3f}", flush=True)
    for unit, name in ((4 << 10, "4 KiB"), (32 << 10, "32 KiB"), (2 << 20, "2 MiB")):
        for which in ("out", "in", "both"):
            row = []
            for k in range(1, 8):
                di = k * unit if which in ("in", "both") else 0
                do = k * unit if which in ("out", "both") else 0
                m = time_at(r, t, base_in + di, base_out + do, rounds=5, warm=3)
                row.append(m[0])
            print(f"{which:4s} + k x {name:6s} k=1..7      " + " ".join(f"{v:.4f}" for v in row), flush=True)
        m = time_at(r, t, base_in, base_out, rounds=5, warm=3)
        print(f"reference again                 median {m[0]:.4f}", flush=True)
    # the output directly behind / far from the input, and the two swapped
    for name, pi, po in (("output right behind the input", base_in, base_in + 8 * N),
                         ("output 1 GiB + 4 KiB behind", base_in, base_in + 8 * N + (1 << 20) + 4096),
                         ("output in front of the input", base_in + 8 * CAP + slack, base_in)):
        if po == base_in:
            api.check(L.sfe_dsp_synth_fill(pi, 2 * N, synth.SEED, 0, 0, None))
        m = time_at(r, t, pi, po, rounds=5, warm=3)
        print(f"{name:31s} median {m[0]:.4f}", flush=True)


def realloc():
    r, t = handle()
    keep = []
    rng = np.random.default_rng(5)
    for i in range(10):
        x = api.DeviceArray(2 * N)
        x.fill_synth(synth.SEED)
        y = api.DeviceArray(2 * CAP)
        m = time_at(r, t, x.ptr, y.ptr, rounds=7, warm=(60 if i == 0 else 4))
        print(f"allocation {i}: in {x.ptr:#014x} out {y.ptr:#014x}  median {m[0]:.4f}  min {m[1]:.4f}  max {m[2]:.4f}  "
              f"({len(keep)} other buffers alive)", flush=True)
        x.free()
        y.free()
        if i % 2 == 1:           # push the next pair elsewhere: odd-sized neighbours stay allocated
            keep.append(api.DeviceArray(int(rng.integers(1 << 24, 1 << 28)) + 1023))


def windows():
    x = api.DeviceArray(2 * N)
    x.fill_synth(synth.SEED)
    y = api.DeviceArray(2 * CAP)
    r, t = handle()
    time_at(r, t, x.ptr, y.ptr, rounds=3, warm=60)
    wins = [1, 2, 4, 8, 16, 64, 256]
    res = {w: [] for w in wins}
    for k in range(8):
        for w in wins:
            os.environ["SFE_TILED_WIN"] = str(w)
            res[w].append(time_at(r, t, x.ptr, y.ptr, rounds=3, warm=2)[0])
    print(f"in {x.ptr:#014x} out {y.ptr:#014x}")
    for w in wins:
        a = np.array(res[w])
        print(f"windows {w:4d}: median {np.median(a):.4f} ms  min {a.min():.4f}  max {a.max():.4f}  frac {frac(float(np.median(a))):.3f}", flush=True)


def tlb():
    """ONE process, diagnostic library: a one-lane look-ahead touch of the tile N tiles further on (address translation
    in flight before that tile's workgroup starts), interleaved A/B against none."""
    x = api.DeviceArray(2 * N)
    x.fill_synth(synth.SEED)
    y = api.DeviceArray(2 * CAP)
    r, t = handle()
    time_at(r, t, x.ptr, y.ptr, rounds=3, warm=60)
    aheads = [0, 256, 1024, 2048, 4096, 16384]
    res = {w: [] for w in aheads}
    for k in range(6):
        for w in aheads:
            os.environ["SFE_TILED_TLB"] = str(w)
            res[w].append(time_at(r, t, x.ptr, y.ptr, rounds=3, warm=2)[0])
    print("ahead:" + "".join(f"  {w}: {np.median(res[w]):.4f}" for w in aheads), flush=True)


def tlbmany(n):
    for i in range(n):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "tlb"], capture_output=True, text=True, timeout=300)
        line = (out.stdout.strip().splitlines() or ["(no output) " + out.stderr[-300:]])[-1]
        print(f"process {i:2d}: {line}", flush=True)


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "one"
    if cmd == "one":
        one()
    elif cmd == "many":
        many(int(sys.argv[2]) if len(sys.argv) > 2 else 20)
    elif cmd == "sweep":
        sweep()
    elif cmd == "realloc":
        realloc()
    elif cmd == "windows":
        windows()
    elif cmd == "tlb":
        tlb()
    elif cmd == "tlbmany":
        tlbmany(int(sys.argv[2]) if len(sys.argv) > 2 else 8)
    else:
        raise SystemExit(__doc__)
