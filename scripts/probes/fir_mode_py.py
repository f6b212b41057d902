"""The FIR's "fast mode"  from a Python process WITHOUT torch: the library through ctypes only, bench.py's order
of events (the pair from sfe_dsp_malloc_pair first, the object after).  One line per process.  Words on the command line pick
the experiment (profiles/r04/fir_modes_input.txt says which block each made): torch, calibrate, two (object or pair, input or
output), pause, slices, pool / pool2 [inputfirst] (the diagnostic library keeps the classified pool), pmc, and zeros -- the one
that settled it: windows of the input read back after the timing (the fast inputs had lost 50-84 % of their data)."""
import ctypes as C
import math
import os
import sys

if "torch" in sys.argv:                          # torch first, as bench.py has it: its bundled HIP runtime is then the process's
    import torch
    torch.cuda.init()
    torch.zeros(16, device="cuda:0")
here = os.path.dirname(os.path.abspath(__file__))
if "pool" in sys.argv or "pool2" in sys.argv:
    os.environ["SFE_PAIR_KEEP_POOL"] = "1"       # (read by libsfe_dsp_diag.so's build_pair)
L = C.CDLL(os.path.join(here, "..", "..", "simplefe_amd", "libsfe_dsp_diag.so" if ("pool" in sys.argv or "pool2" in sys.argv) else "libsfe_dsp.so"))
L.sfe_dsp_last_error.restype = C.c_char_p


def ck(rc):
    if rc != 0:
        raise RuntimeError(L.sfe_dsp_last_error().decode())


n = 1 << 28
if "pool2" in sys.argv:
    # what is done to the input's chunks that is not done to the rest of the pool?  Slots of the kept pool treated one way each
    # BEFORE the pair's own input is touched: A = the first buffer the FIR ever runs on; B = read by the bare mix first (as the
    # library's verification reads the input); C = nothing special; then the input itself
    din, dout = C.c_void_p(), C.c_void_p()
    kept, worst = C.c_float(), C.c_float()
    ck(L.sfe_dsp_malloc_pair(C.c_size_t(n * 8), C.c_size_t(n * 8), 4, C.byref(din), C.byref(dout), C.byref(kept), C.byref(worst)))
    pool, pn = C.c_void_p(), C.c_size_t()
    ck(L.sfe_dsp_diag_last_pool(C.byref(pool), C.byref(pn)))
    if not pool.value or pn.value < 40:
        print("pool2: no pool kept (a pair of plain allocations)")
        sys.exit(0)
    GiB = 1 << 30
    A, B, Cc = (C.c_void_p(pool.value + k * GiB) for k in (10, 20, 30))
    taps = (C.c_float * 256)()
    for i in range(256):
        x = 0.2 * (i - 127.5)
        taps[i] = math.sin(math.pi * x) / (math.pi * x) * (0.54 - 0.46 * math.cos(2 * math.pi * i / 255.0)) * 0.2
    f = C.c_void_p()
    tm = C.c_void_p()
    ck(L.sfe_dsp_timer_create(C.byref(tm)))

    def fill(a):
        ck(L.sfe_dsp_synth_fill(a, C.c_uint64(2 * n), C.c_uint32(20240601), C.c_uint32(0), C.c_uint64(0), None))

    def run(a, warm=10, reps=30):
        for _ in range(warm):
            ck(L.sfe_dsp_fir_process_stream(f, a, dout, C.c_size_t(n), C.c_size_t(n), C.c_size_t(n), None))
        ck(L.sfe_dsp_timer_start(tm, None))
        for _ in range(reps):
            ck(L.sfe_dsp_fir_process_stream(f, a, dout, C.c_size_t(n), C.c_size_t(n), C.c_size_t(n), None))
        ck(L.sfe_dsp_timer_stop(tm, None))
        ms = C.c_float()
        ck(L.sfe_dsp_timer_elapsed_ms(tm, C.byref(ms)))
        return ms.value / reps
    if "inputfirst" in sys.argv:
        # the pair's input first (as bench.py has it), then a control slot, then a slot read by the bare mix before it is filled
        fill(din)
        ck(L.sfe_dsp_fir_create(taps, 256, 0, 1, 1, 0, 0, C.byref(f)))
        tI = run(din, 40, 50)
        fill(A)
        tA = run(A)
        mix = C.c_float()
        for _ in range(2):
            ck(L.sfe_dsp_probe_pair(B, C.c_size_t(n * 8), dout, C.c_size_t(n * 8), C.byref(mix)))
        fill(B)
        tB = run(B)
        # and a slot whose chunks are first read by the FIR itself before they are filled (zeros or the classification's leavings)
        tC0 = run(Cc)
        fill(Cc)
        tC = run(Cc)
        print("pool2 inputfirst: the pair's input %.4f  A (control) %.4f  B (read by the bare mix, then filled) %.4f  C (read by the FIR unfilled %.4f, then filled) %.4f | again: %.4f %.4f %.4f %.4f" % (
            tI, tA, tB, tC0, tC, run(din), run(A), run(B), run(Cc)))
        sys.exit(0)
    fill(A)
    ck(L.sfe_dsp_fir_create(taps, 256, 0, 1, 1, 0, 0, C.byref(f)))
    tA = run(A, 200, 50)
    mix = C.c_float()
    for _ in range(2):
        ck(L.sfe_dsp_probe_pair(B, C.c_size_t(n * 8), dout, C.c_size_t(n * 8), C.byref(mix)))
    fill(B)
    tB = run(B)
    fill(Cc)
    tC = run(Cc)
    fill(din)
    tI = run(din, 40, 50)
    print("pool2: A (the FIR's first buffer) %.4f  B (read by the bare mix first) %.4f  C (control) %.4f  the pair's input %.4f | again: %.4f %.4f %.4f %.4f" % (
        tA, tB, tC, tI, run(A), run(B), run(Cc), run(din)))
    sys.exit(0)
if "slices" in sys.argv:
    # the first pair of the process with a 16 GiB input: eight 2 GiB slices of it (two chunks each, all of the input's class) as the
    # FIR's input, one output.  All eight alike (the pool's or the process's property) or mixed (the chunks')?  (Tried once, eight
    # processes: the pool did not hold the 18 chunks of one class this needs, the pairs were plain allocations, all slow.)
    din, dout = C.c_void_p(), C.c_void_p()
    kept, worst = C.c_float(), C.c_float()
    ck(L.sfe_dsp_malloc_pair(C.c_size_t(8 * n * 8), C.c_size_t(n * 8), 4, C.byref(din), C.byref(dout), C.byref(kept), C.byref(worst)))
    kind = C.c_int()
    ck(L.sfe_dsp_mem_kind(din, C.byref(kind)))
    ck(L.sfe_dsp_synth_fill(din, C.c_uint64(16 * n), C.c_uint32(20240601), C.c_uint32(0), C.c_uint64(0), None))
    taps = (C.c_float * 256)()
    for i in range(256):
        x = 0.2 * (i - 127.5)
        taps[i] = math.sin(math.pi * x) / (math.pi * x) * (0.54 - 0.46 * math.cos(2 * math.pi * i / 255.0)) * 0.2
    f = C.c_void_p()
    ck(L.sfe_dsp_fir_create(taps, 256, 0, 1, 1, 0, 0, C.byref(f)))
    tm = C.c_void_p()
    ck(L.sfe_dsp_timer_create(C.byref(tm)))
    res = []
    for rnd in range(2):
        for k in range(8):
            a = C.c_void_p(din.value + k * n * 8)
            for _ in range(40 if (rnd == 0 and k == 0) else 10):
                ck(L.sfe_dsp_fir_process_stream(f, a, dout, C.c_size_t(n), C.c_size_t(n), C.c_size_t(n), None))
            ck(L.sfe_dsp_timer_start(tm, None))
            for _ in range(30):
                ck(L.sfe_dsp_fir_process_stream(f, a, dout, C.c_size_t(n), C.c_size_t(n), C.c_size_t(n), None))
            ck(L.sfe_dsp_timer_stop(tm, None))
            ms = C.c_float()
            ck(L.sfe_dsp_timer_elapsed_ms(tm, C.byref(ms)))
            res.append(ms.value / 30)
    print("slices of a 16 GiB first input (%s): %s | again: %s" % ("built from chunks" if kind.value else "plain allocation", " ".join("%.4f" % v for v in res[:8]), " ".join("%.4f" % v for v in res[8:])))
    sys.exit(0)
din, dout = C.c_void_p(), C.c_void_p()
kept, worst = C.c_float(), C.c_float()
ck(L.sfe_dsp_malloc_pair(C.c_size_t(n * 8), C.c_size_t(n * 8), 4, C.byref(din), C.byref(dout), C.byref(kept), C.byref(worst)))
ck(L.sfe_dsp_synth_fill(din, C.c_uint64(2 * n), C.c_uint32(20240601), C.c_uint32(0), C.c_uint64(0), None))
taps = (C.c_float * 256)()
for i in range(256):
    x = 0.2 * (i - 127.5)
    taps[i] = math.sin(math.pi * x) / (math.pi * x) * (0.54 - 0.46 * math.cos(2 * math.pi * i / 255.0)) * 0.2
f = C.c_void_p()
ck(L.sfe_dsp_fir_create(taps, 256, 0, 1, 1, 0, 0, C.byref(f)))
tm = C.c_void_p()
ck(L.sfe_dsp_timer_create(C.byref(tm)))
if "calibrate" in sys.argv:
    ck(L.sfe_dsp_fir_calibrate(f, din, dout, C.c_size_t(n), C.c_size_t(n), C.c_size_t(n), None, None))
for _ in range(8 if "pmc" in sys.argv else 40):
    ck(L.sfe_dsp_fir_process_stream(f, din, dout, C.c_size_t(n), C.c_size_t(n), C.c_size_t(n), None))
out = []
for r in range(3):
    ck(L.sfe_dsp_timer_start(tm, None))
    for _ in range(4 if "pmc" in sys.argv else 50):
        ck(L.sfe_dsp_fir_process_stream(f, din, dout, C.c_size_t(n), C.c_size_t(n), C.c_size_t(n), None))
    ck(L.sfe_dsp_timer_stop(tm, None))
    ms = C.c_float()
    ck(L.sfe_dsp_timer_elapsed_ms(tm, C.byref(ms)))
    out.append(ms.value / (4 if "pmc" in sys.argv else 50))
if "pmc" in sys.argv:
    # under rocprofv3 --pmc: a second pair, then the LAST 16 launches of the process are 4 x (in1 -> out1), 4 x (in2 -> out2),
    # 4 x (in1 -> out2), 4 x (in2 -> out1): scripts/probes/fir_mode_pmc.sh reads their counters by position
    din2, dout2 = C.c_void_p(), C.c_void_p()
    ck(L.sfe_dsp_malloc_pair(C.c_size_t(n * 8), C.c_size_t(n * 8), 4, C.byref(din2), C.byref(dout2), None, None))
    ck(L.sfe_dsp_synth_fill(din2, C.c_uint64(2 * n), C.c_uint32(20240601), C.c_uint32(0), C.c_uint64(0), None))
    res = []
    for a, b in ((din, dout), (din2, dout2), (din, dout2), (din2, dout)):
        ck(L.sfe_dsp_timer_start(tm, None))
        for _ in range(4):
            ck(L.sfe_dsp_fir_process_stream(f, a, b, C.c_size_t(n), C.c_size_t(n), C.c_size_t(n), None))
        ck(L.sfe_dsp_timer_stop(tm, None))
        ms = C.c_float()
        ck(L.sfe_dsp_timer_elapsed_ms(tm, C.byref(ms)))
        res.append(ms.value / 4)
    print("pmc order: in1->out1 %.4f  in2->out2 %.4f  in1->out2 %.4f  in2->out1 %.4f (events around 4 launches, profiler attached)" % tuple(res))
    sys.exit(0)
if "zeros" in sys.argv:
    # IS THE FAST INPUT STILL HOLDING ITS DATA?  (The FIR on a buffer of zeros runs at the fast mode's speed: block 19.)  Windows of
    # the input read back after the timing; then the input filled AGAIN and timed again
    import time
    hip = C.CDLL("libamdhip64.so.7")

    def zero_fraction(ptr, nbytes, windows=512, wlen=4096):
        buf = (C.c_ubyte * wlen)()
        z = 0
        for k in range(windows):
            off = (nbytes - wlen) * k // (windows - 1)
            off -= off % 8
            rc = hip.hipMemcpy(buf, C.c_void_p(ptr.value + off), C.c_size_t(wlen), 2)
            if rc != 0:
                raise RuntimeError("hipMemcpy %d" % rc)
            z += 1 if not any(buf) else 0
        return z / windows

    def run(a, b):
        for _ in range(10):
            ck(L.sfe_dsp_fir_process_stream(f, a, b, C.c_size_t(n), C.c_size_t(n), C.c_size_t(n), None))
        ck(L.sfe_dsp_timer_start(tm, None))
        for _ in range(40):
            ck(L.sfe_dsp_fir_process_stream(f, a, b, C.c_size_t(n), C.c_size_t(n), C.c_size_t(n), None))
        ck(L.sfe_dsp_timer_stop(tm, None))
        ms = C.c_float()
        ck(L.sfe_dsp_timer_elapsed_ms(tm, C.byref(ms)))
        return ms.value / 40
    t1, z1 = run(din, dout), zero_fraction(din, n * 8)
    ck(L.sfe_dsp_synth_fill(din, C.c_uint64(2 * n), C.c_uint32(20240601), C.c_uint32(0), C.c_uint64(0), None))
    t2, z2 = run(din, dout), zero_fraction(din, n * 8)
    time.sleep(2.0)
    t3, z3 = run(din, dout), zero_fraction(din, n * 8)
    print("  as first filled: FIR %.4f ms, %.0f %% of 512 windows of the input read back all zero | filled again: %.4f ms, %.0f %% zero | 2 s later: %.4f ms, %.0f %% zero" % (
        t1, 100 * z1, t2, 100 * z2, t3, 100 * z3))
if "pool" in sys.argv:
    # the diagnostic library keeps the REST of the classified pool mapped (64 chunks in creation order, without the four the pair
    # took).  The first pair timed with the pool held; every chunk of the pool as the FIR's input (2^27 samples, output = the
    # pair's); the input's own two chunks the same way; the pool released; the pair timed again.
    def run(a, b, cnt, warm=10, reps=20):
        for _ in range(warm):
            ck(L.sfe_dsp_fir_process_stream(f, a, b, C.c_size_t(cnt), C.c_size_t(cnt), C.c_size_t(cnt), None))
        ck(L.sfe_dsp_timer_start(tm, None))
        for _ in range(reps):
            ck(L.sfe_dsp_fir_process_stream(f, a, b, C.c_size_t(cnt), C.c_size_t(cnt), C.c_size_t(cnt), None))
        ck(L.sfe_dsp_timer_stop(tm, None))
        ms = C.c_float()
        ck(L.sfe_dsp_timer_elapsed_ms(tm, C.byref(ms)))
        return ms.value / reps
    pool, pn = C.c_void_p(), C.c_size_t()
    ck(L.sfe_dsp_diag_last_pool(C.byref(pool), C.byref(pn)))
    GiB = 1 << 30
    held = run(din, dout, n, 10, 40)
    line = "  pair, pool of %d held: %.4f" % (pn.value, held)
    half = n // 2
    line += " | the input's two chunks alone (2^27): %.4f %.4f" % (run(din, dout, half), run(C.c_void_p(din.value + GiB), dout, half))
    if pool.value and pn.value:
        ck(L.sfe_dsp_synth_fill(pool, C.c_uint64(pn.value * GiB // 4), C.c_uint32(20240601), C.c_uint32(0), C.c_uint64(0), None))
        per = [run(C.c_void_p(pool.value + c * GiB), dout, half, 5, 12) for c in range(pn.value)]
        line += " | pool chunks as input (2^27): " + " ".join("%.3f" % v for v in per)
        two = [run(C.c_void_p(pool.value + c * GiB), dout, n, 5, 12) for c in range(0, pn.value - 1, 2)]
        line += " | pool slots of two chunks as input (2^28): " + " ".join("%.3f" % v for v in two)
        ck(L.sfe_dsp_free(pool))
        line += " | pool released, the pair again: %.4f" % run(din, dout, n, 10, 40)
    print(line)
if "pause" in sys.argv:
    # inside ONE process: the first pair freed, a pause, a second pair; that freed, no pause, a third.  Does the pause that moves
    # the odds between processes (profiles/r04/fir_modes_input.txt, block 13) do the same after a release inside a process?
    import time

    def run(a, b):
        for _ in range(20):
            ck(L.sfe_dsp_fir_process_stream(f, a, b, C.c_size_t(n), C.c_size_t(n), C.c_size_t(n), None))
        ck(L.sfe_dsp_timer_start(tm, None))
        for _ in range(40):
            ck(L.sfe_dsp_fir_process_stream(f, a, b, C.c_size_t(n), C.c_size_t(n), C.c_size_t(n), None))
        ck(L.sfe_dsp_timer_stop(tm, None))
        ms = C.c_float()
        ck(L.sfe_dsp_timer_elapsed_ms(tm, C.byref(ms)))
        return ms.value / 40
    line = "  first pair %.4f" % run(din, dout)
    pause = float(sys.argv[sys.argv.index("pause") + 1]) if len(sys.argv) > sys.argv.index("pause") + 1 else 5.0
    for wait in (pause, 0.0, pause, 0.0):
        ck(L.sfe_dsp_free(din))
        ck(L.sfe_dsp_free(dout))
        time.sleep(wait)
        din, dout = C.c_void_p(), C.c_void_p()
        ck(L.sfe_dsp_malloc_pair(C.c_size_t(n * 8), C.c_size_t(n * 8), 4, C.byref(din), C.byref(dout), None, None))
        ck(L.sfe_dsp_synth_fill(din, C.c_uint64(2 * n), C.c_uint32(20240601), C.c_uint32(0), C.c_uint64(0), None))
        line += " | freed, %.0f s, built again: %.4f" % (wait, run(din, dout))
    print(line)
    sys.exit(0)
if "two" in sys.argv:
    # does the mode follow the OBJECT (its spectrum, twiddle tables and ticket counters: small allocations made after the pool
    # went back) or the PAIR?  a second object, made now; a second pair, built now; all four combinations
    def run(obj, a, b):
        for _ in range(10):
            ck(L.sfe_dsp_fir_process_stream(obj, a, b, C.c_size_t(n), C.c_size_t(n), C.c_size_t(n), None))
        ck(L.sfe_dsp_timer_start(tm, None))
        for _ in range(50):
            ck(L.sfe_dsp_fir_process_stream(obj, a, b, C.c_size_t(n), C.c_size_t(n), C.c_size_t(n), None))
        ck(L.sfe_dsp_timer_stop(tm, None))
        ms = C.c_float()
        ck(L.sfe_dsp_timer_elapsed_ms(tm, C.byref(ms)))
        return ms.value / 50
    f2 = C.c_void_p()
    ck(L.sfe_dsp_fir_create(taps, 256, 0, 1, 1, 0, 0, C.byref(f2)))
    din2, dout2 = C.c_void_p(), C.c_void_p()
    ck(L.sfe_dsp_malloc_pair(C.c_size_t(n * 8), C.c_size_t(n * 8), 4, C.byref(din2), C.byref(dout2), None, None))
    ck(L.sfe_dsp_synth_fill(din2, C.c_uint64(2 * n), C.c_uint32(20240601), C.c_uint32(0), C.c_uint64(0), None))
    f3 = C.c_void_p()
    ck(L.sfe_dsp_fir_create(taps, 256, 0, 1, 1, 0, 0, C.byref(f3)))
    print("  object 1 / 2 / 3 on pair 1: %.4f %.4f %.4f   on pair 2: %.4f %.4f %.4f   crossed (in 1 -> out 2, in 2 -> out 1), object 1: %.4f %.4f" % (
        run(f, din, dout), run(f2, din, dout), run(f3, din, dout), run(f, din2, dout2), run(f2, din2, dout2), run(f3, din2, dout2),
        run(f, din, dout2), run(f, din2, dout)))
print("python %s: FIR %.4f %.4f %.4f ms   bare mix %.4f (own class %.4f)" % (
    " + ".join(sys.argv[1:]) or "(ctypes only)", out[0], out[1], out[2], kept.value, worst.value))
