#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF (run in the authoring container).

  resample / decimate : outputs of the unmodified /root/reference/libdsp/{resample,decimate}.cxx
      compiled in place by oracle/Makefile into oracle/_ref/libsferef.so (-O2 -ffp-contract=off).
  blkconv             : the reference class itself needs a GPU box to run (it is built on ROCm's
      libhipfftw): those fixtures are g6_*.npz, made by make_golden_blkconv.py.  The ones made
      here are (a) the known-answer scenario of
      libdsp/test/test_blkconv.cxx:5-33 -- 5-tap boxcar, fft 32, a block of ones then a block
      of zeros -- with the values that program prints (to %.2f), and (b) float64 direct linear
      convolution, the mathematical definition blkconv.cxx:77-110 implements.

The fixtures are data (inputs + expected outputs); no reference source text is stored.
The 31 test taps are read, as numbers, out of libdsp/test/test_decimate.py:13 at generation
time (they are the reference test's input vector).

    python tests/golden/make_golden.py        # rewrites the .npz files next to this script
"""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import binding as orc  # noqa: E402
from simplefe_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def reference_test_taps():
    txt = open(os.path.join(REF, "libdsp/test/test_decimate.py")).read()
    m = re.search(r"^taps\s*=\s*\[([^\]]*)\]", txt, re.M)
    vals = [float(v) for v in m.group(1).split(",")]
    assert len(vals) == 31
    return np.array(vals, dtype=np.float32)   # SWIG IN_ARRAY1 float* -> float32 (pydsp.i:16)


def run(cls, taps, U, B, x, rate, out_len):
    o = cls(taps, U, B)
    y, ns = o.stream(x, rate, chunk=B, out_len=out_len)
    return y, np.array(ns, dtype=np.int32)


def main():
    orc.build(ref=True)
    assert orc.ref_lib() is not None, "reference build failed"

    # ---- G4: the reference's own test vector (test_decimate.py / test_resample.py) -------
    taps31 = reference_test_taps()
    N, B, U = 1024, 128, 4
    x0 = np.sin(0.02 * np.pi * np.arange(N)).astype(np.float32)   # test_decimate.py:16
    g4 = {"taps": taps31, "x": x0, "U": U, "B": B}
    for tag, rate in (("1p77", 1.77), ("5o3", 5.0 / 3.0), ("8", 8.0), ("2p5", 2.5)):
        yr, nr = run(orc.RefResample, taps31, U, B, x0, rate, 4 * B)
        yd, nd = run(orc.RefDecimate, taps31, U, B, x0, rate, 4 * B)
        assert np.array_equal(yr, yd) and np.array_equal(nr, nd), tag   # test_decimate.py:36
        g4[f"rate_{tag}"] = np.float32(rate)
        g4[f"y_{tag}"] = yr
        g4[f"n_{tag}"] = nr
        print(f"G4 rate {rate:.4f}: n_out {len(yr)}  y[0:4] {yr[:4]}")
    yr, nr = run(orc.RefResample, taps31, U, B, x0, 0.77, 4 * B)       # test_resample.py:24
    g4["rate_0p77"] = np.float32(0.77)
    g4["y_0p77"] = yr
    g4["n_0p77"] = nr
    print(f"G4 rate 0.77: n_out {len(yr)} peak {np.abs(yr).max():.4f}")
    np.savez_compressed(os.path.join(OUT, "g4_reference_test_vector.npz"), **g4)

    # ---- G5: BASELINE cfg3 / cfg4 shapes on 2^16-sample seeded streams ------------------
    n = 1 << 16
    x = synth.synth_f32(n, seed=synth.SEED, ch=0)
    g5 = {"n": n, "seed": synth.SEED}
    for name, taps, U, rate in (("cfg3", synth.taps_cfg3(), 3, 5.0 / 3.0),
                                ("cfg4", synth.taps_cfg4(), 1, 8.0),
                                ("gen", taps31, 4, 1.77),
                                ("gen2", synth.taps_cfg3(), 3, 1.3)):
        g5[f"{name}_taps"] = taps
        g5[f"{name}_U"] = U
        g5[f"{name}_rate"] = np.float32(rate)
        first = None
        for B in (4096, 1000, 1001):
            ol = int(np.ceil(B / rate)) + 2
            yr, nr = run(orc.RefResample, taps, U, B, x, rate, ol)
            yd, nd = run(orc.RefDecimate, taps, U, B, x, rate, ol)
            assert np.array_equal(yr, yd) and np.array_equal(nr, nd), (name, B)
            g5[f"{name}_n_B{B}"] = nr
            step = np.float32(rate) * np.float32(U)
            if float(step) == np.floor(float(step)):
                # integer-valued step: chunking must not matter
                if first is None:
                    first = yr
                    g5[f"{name}_y"] = yr
                assert np.array_equal(first, yr), (name, B)
            else:
                g5[f"{name}_y_B{B}"] = yr
            print(f"G5 {name} B={B}: n_out {len(yr)}")
    np.savez_compressed(os.path.join(OUT, "g5_baseline_shapes.npz"), **g5)

    # ---- G1/G2: blkconv ------------------------------------------------------------
    g1 = {
        "taps": np.ones(5, dtype=np.float32), "fft_len": 32, "blksize": 28,      # test_blkconv.cxx:7-12
        "in1": np.ones(28, dtype=np.float32), "in2": np.zeros(28, dtype=np.float32),
        # values test_blkconv.cxx:19-31 prints with "%.2f"
        "out1": np.array([1, 2, 3, 4] + [5] * 24, dtype=np.float32),
        "out2": np.array([4, 3, 2, 1] + [0] * 24, dtype=np.float32),
        "print_tol": np.float32(0.005),
    }
    # G2: pulse-shaping use (examples/bpsk/bpsk.cxx:122-164 pattern: +-0.6296 impulses every
    # 10 samples through a 111-tap filter, fft 2048); taps are an own RRC-like prototype.
    rng = np.random.default_rng(7)
    k = np.arange(111) - 55
    beta, sps = 0.35, 10.0
    t = k / sps
    with np.errstate(divide="ignore", invalid="ignore"):
        h = (np.sin(np.pi * t * (1 - beta)) + 4 * beta * t * np.cos(np.pi * t * (1 + beta))) / (
            np.pi * t * (1 - (4 * beta * t) ** 2))
    h[55] = 1 - beta + 4 * beta / np.pi
    bad = ~np.isfinite(h)
    h[bad] = 0.0
    h = (h / np.sqrt(np.sum(h * h))).astype(np.float32)
    blk = 2048 + 1 - 111
    nblk = 8
    xs = np.zeros(blk * nblk, dtype=np.float32)
    bits = rng.integers(0, 2, size=len(xs) // 10 + 1)
    xs[::10] = np.where(bits[: len(xs[::10])] > 0, -0.85 / 1.35, 0.85 / 1.35).astype(np.float32)
    y64 = np.convolve(xs.astype(np.float64), h.astype(np.float64))[: len(xs)]
    g1.update({"g2_taps": h, "g2_fft_len": 2048, "g2_x": xs, "g2_y64": y64})
    np.savez_compressed(os.path.join(OUT, "g1_blkconv.npz"), **g1)
    print("G1/G2 blkconv written")


if __name__ == "__main__":
    main()
