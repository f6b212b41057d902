#!/usr/bin/env python3
"""Generate tests/golden/g7_blkconv_fftw.npz: outputs of the REFERENCE ITSELF, run here.

The reference's blkconv class (/root/reference/libdsp/blkconv.cxx, unmodified) calling the
reference's own FFTW 3.3.5 single-precision library (/root/reference/contrib/fftw-3.3.5-dll64/
libfftw3f-3.dll, unmodified, read where it lies): the DLL is x86-64 code, oracle/pe/ maps it into
this process (`make -C oracle ref_fftw`).  CPU only, authoring container only; the fixture holds
inputs and the reference's outputs (data only) and is what travels.

    python tests/golden/make_golden_fftw.py            # writes tests/golden/g7_blkconv_fftw.npz

Cases (the g6 inputs, so FFTW / hipFFTW / the port can be tabulated against each other, plus one):
  kat   the reference test program's scenario, libdsp/test/test_blkconv.cxx:5-33
  bpsk  the pulse-shaping stream of examples/bpsk/bpsk.cxx:122-164 (111 taps, fft 2048)
  cfg1  BASELINE configs[0]: 63 taps, fft 1024
  cfg2  BASELINE configs[1]: 256 taps, fft 4096
  rrc551  the long-prototype shape of examples/bpsk/bpsk.cxx:58-63 (551 taps, fft 8192); the taps
        are an own root-raised-cosine by formula (roll-off 0.35, 10 samples per symbol), not the
        reference's table
Every case is fed block by block through get_process_buf() / process().
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import binding as orc  # noqa: E402
from simplefe_amd import synth  # noqa: E402


def rrc_taps(n_taps=551, sps=10, beta=0.35):
    """Root-raised-cosine prototype, unit energy, float64 formula rounded to float32."""
    t = (np.arange(n_taps, dtype=np.float64) - (n_taps - 1) / 2) / sps
    h = np.empty_like(t)
    for i, ti in enumerate(t):
        if abs(ti) < 1e-12:
            h[i] = 1.0 - beta + 4 * beta / np.pi
        elif abs(abs(4 * beta * ti) - 1.0) < 1e-9:
            h[i] = beta / np.sqrt(2) * ((1 + 2 / np.pi) * np.sin(np.pi / (4 * beta)) + (1 - 2 / np.pi) * np.cos(np.pi / (4 * beta)))
        else:
            h[i] = (np.sin(np.pi * ti * (1 - beta)) + 4 * beta * ti * np.cos(np.pi * ti * (1 + beta))) / (np.pi * ti * (1 - (4 * beta * ti) ** 2))
    h /= np.sqrt(np.sum(h * h))
    return h.astype(np.float32)


def cases():
    g6 = np.load(os.path.join(ROOT, "tests", "golden", "g6_blkconv_reference.npz"))
    for name in ("kat", "bpsk", "cfg1", "cfg2"):
        yield name, g6[f"{name}_taps"], int(g6[f"{name}_fft_len"]), g6[f"{name}_x"], g6[f"{name}_y"]
    taps = rrc_taps()
    blk = 8192 + 1 - len(taps)
    yield "rrc551", taps, 8192, synth.synth_f32(4 * blk, ch=551), None


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "g7_blkconv_fftw.npz")
    ver = orc.RefBlkconvFFTW.fftw_version()
    if ver is None:
        raise SystemExit("the FFTW-pinned reference is not usable here (no /root/reference, or SFE_ORACLE_RUN_FFTW_DLL=1 not set)")
    print("FFTW binary reports:", ver)
    g = {}
    print(f"{'case':8s} {'taps':>5s} {'fft':>5s} {'port~FFTW':>10s} {'hipFFTW~FFTW':>13s} {'FFTW~f64':>10s} {'port~f64':>10s} {'hipFFTW~f64':>12s}")
    for name, taps, fft_len, x, y_hip in cases():
        r = orc.RefBlkconvFFTW(taps, fft_len)       # (round 5: runs in a child process, opt-in: SFE_ORACLE_RUN_FFTW_DLL=1)
        assert len(x) % (fft_len + 1 - len(taps)) == 0
        y = r.stream_blocks(x)                      # block by block through the process buffer
        assert np.array_equal(y, orc.RefBlkconvFFTW(taps, fft_len).stream(x))
        g.update({f"{name}_taps": taps, f"{name}_fft_len": fft_len, f"{name}_x": x, f"{name}_y": y})
        o = orc.Blkconv(taps, fft_len).stream(x)
        d = np.convolve(x.astype(np.float64), taps.astype(np.float64))[:len(x)]
        hip = f"{synth.rel_rms(y_hip, y):13.2e}" if y_hip is not None else f"{'-':>13s}"
        hipd = f"{synth.rel_rms(y_hip, d):12.2e}" if y_hip is not None else f"{'-':>12s}"
        print(f"{name:8s} {len(taps):5d} {fft_len:5d} {synth.rel_rms(o, y):10.2e} {hip} {synth.rel_rms(y, d):10.2e} "
              f"{synth.rel_rms(o, d):10.2e} {hipd}", flush=True)
    print("kat:", g["kat_y"][:6], g["kat_y"][26:34])
    np.savez_compressed(out_path, **g)
    print("wrote", out_path, os.path.getsize(out_path), "bytes")


if __name__ == "__main__":
    main()
