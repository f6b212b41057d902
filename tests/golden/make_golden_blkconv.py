#!/usr/bin/env python3
"""Generate tests/golden/g6_blkconv_reference.npz from the REFERENCE's own blkconv class.

The class (libdsp/blkconv.cxx, unmodified) is compiled in the authoring container by
oracle/Makefile against the reference's vendored fftw3.h and ROCm's libhipfftw.so into
oracle/_ref/libsferef_blkconv.so.  hipFFT executes on the device, so THIS SCRIPT RUNS ON A GPU
BOX (it reads nothing under /root/reference -- only the prebuilt .so that travelled with the
snapshot):

    gpurun -- 'python tests/golden/make_golden_blkconv.py gpurun_out/g6_blkconv_reference.npz'
    cp gpurun_out/g6_blkconv_reference.npz tests/golden/

Cases: the reference test program's scenario (test_blkconv.cxx:5-33: 5-tap boxcar, fft 32, a
block of ones then a block of zeros), the bpsk pulse-shaping stream of fixture G2 (111 taps, fft
2048, +-0.63 impulses every 10 samples: examples/bpsk/bpsk.cxx:122-164), BASELINE cfg1 (63 taps, fft 1024) and cfg2 (256 taps,
fft 4096), each fed several blocks of the synthetic stream through get_process_buf()/process().
The fixture holds inputs and the reference's outputs (data only).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import binding as orc  # noqa: E402
from simplefe_amd import synth  # noqa: E402


def run_blocks(taps, fft_len, x):
    r = orc.RefBlkconv(taps, fft_len)
    assert r.blk == fft_len + 1 - len(taps)
    assert len(x) % r.blk == 0
    return r.stream(x), r.blk


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "g6_blkconv_reference.npz")
    g = {}
    # the reference test program's scenario
    taps = np.ones(5, np.float32)
    x = np.concatenate([np.ones(28, np.float32), np.zeros(28, np.float32)])
    y, blk = run_blocks(taps, 32, x)
    g.update(kat_taps=taps, kat_fft_len=32, kat_x=x, kat_y=y)
    g1 = np.load(os.path.join(ROOT, "tests", "golden", "g1_blkconv.npz"))   # the bpsk-pattern stream of G2
    for name, taps, fft_len, nblk in (("bpsk", g1["g2_taps"], int(g1["g2_fft_len"]), 8),
                                      ("cfg1", synth.taps_cfg1(), 1024, 9),
                                      ("cfg2", synth.taps_cfg2(), 4096, 5)):
        blk = fft_len + 1 - len(taps)
        x = g1["g2_x"] if name == "bpsk" else synth.synth_f32(nblk * blk, ch=len(taps))
        assert len(x) == nblk * blk
        y, _ = run_blocks(taps, fft_len, x)
        g.update({f"{name}_taps": taps, f"{name}_fft_len": fft_len, f"{name}_x": x, f"{name}_y": y})
        o = orc.Blkconv(taps, fft_len).stream(x)
        print(f"{name}: {len(taps)} taps fft {fft_len}: restatement vs reference rel-RMS {synth.rel_rms(o, y):.3e}, "
              f"max abs {np.abs(o - y).max():.3e}", flush=True)
    o = orc.Blkconv(g["kat_taps"], 32).stream(g["kat_x"])
    print("kat: max abs", np.abs(o - g["kat_y"]).max(), g["kat_y"][:8], g["kat_y"][26:34])
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    np.savez_compressed(out_path, **g)
    print("wrote", out_path)


if __name__ == "__main__":
    main()
