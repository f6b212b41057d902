"""Deterministic synthetic I/Q streams and filter prototypes.

The sample generator is a counter-based 32-bit hash, defined once here and implemented
identically on the device (csrc/synth.hip: sfe_dsp_synth_fill), so a 2^30-sample stream
can be generated in HBM and any window of it reproduced on the host for checking.
Values are (int32(hash) >> 8) * 2^-23: exactly representable float32 in [-1, 1).
"""
import numpy as np

SEED = 20240601

_M32 = np.uint64(0xFFFFFFFF)


def hash32(seed, ch, idx):
    """lowbias32-style finaliser over (seed, channel, 64-bit float index)."""
    idx = np.asarray(idx, dtype=np.uint64)
    lo = idx & _M32
    hi = idx >> np.uint64(32)
    x = (lo * np.uint64(0x9E3779B9) + hi * np.uint64(0x7F4A7C15)
         + np.uint64(int(seed) & 0xFFFFFFFF) * np.uint64(0x85EBCA6B)
         + np.uint64(int(ch) & 0xFFFFFFFF) * np.uint64(0xC2B2AE35)
         + np.uint64(0x165667B1)) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def synth_f32(n_floats, seed=SEED, ch=0, first=0):
    """n_floats float32 values; float index i of channel ch is hash32(seed, ch, first+i)."""
    idx = np.arange(first, first + n_floats, dtype=np.uint64)
    u = hash32(seed, ch, idx)
    return (u.view(np.int32) >> 8).astype(np.float32) * np.float32(2.0 ** -23)


def synth_cf32(n_samples, seed=SEED, ch=0, first_sample=0):
    """Interleaved (re, im) float32 array of 2*n_samples floats (gr_complex layout,
    gr-simplefe/lib/source_c_impl.cc:46)."""
    return synth_f32(2 * n_samples, seed, ch, 2 * first_sample)


def lowpass_taps(n_taps, cutoff, gain=1.0):
    """Hamming-windowed sinc, computed in float64 and rounded to float32.
    cutoff is a fraction of Nyquist; DC gain is `gain`."""
    k = np.arange(n_taps, dtype=np.float64) - (n_taps - 1) / 2.0
    h = np.sinc(cutoff * k) * (0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(n_taps) / max(n_taps - 1, 1)))
    h *= gain / h.sum()
    return h.astype(np.float32)


def complex_taps(n_taps, cutoff, shift=0.1):
    """The same prototype shifted by e^{j*shift*pi*k}: returns (re, im) float32 arrays."""
    h = lowpass_taps(n_taps, cutoff).astype(np.float64)
    k = np.arange(n_taps)
    w = np.exp(1j * shift * np.pi * k)
    return (h * w.real).astype(np.float32), (h * w.imag).astype(np.float32)


# BASELINE.json configs
def taps_cfg1():
    return lowpass_taps(63, 0.25)


def taps_cfg2():
    return lowpass_taps(256, 0.2)


def taps_per_channel(n_channels, n_taps=256):
    """cfg5's "per-channel-distinct taps" variant: channel c's 256-tap Hamming
    windowed-sinc low-pass at cut-off 0.10 + 0.30 c / n_channels (float64 formula, rounded to float32)."""
    k = np.arange(n_taps, dtype=np.float64) - (n_taps - 1) / 2.0
    w = 0.54 - 0.46 * np.cos(2.0 * np.pi * np.arange(n_taps) / (n_taps - 1))
    out = np.empty((n_channels, n_taps), dtype=np.float32)
    for c in range(n_channels):
        fc = 0.10 + 0.30 * c / n_channels
        h = fc * np.sinc(fc * k) * w
        out[c] = (h / h.sum()).astype(np.float32)
    return out


def taps_cfg3():
    """381-tap prototype for U=3 (127 taps per polyphase arm), rate 5/3, DC gain U."""
    return lowpass_taps(381, 0.18, gain=3.0)


def taps_cfg3_short():
    """The other reading of configs[2]: a 127-tap PROTOTYPE for U=3 (43 taps per
    polyphase arm), same cutoff and gain."""
    return lowpass_taps(127, 0.18, gain=3.0)


def taps_cfg4():
    return lowpass_taps(64, 0.9 / 8.0)


def rel_rms(y, ref):
    y = np.asarray(y, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    den = np.sqrt(np.sum(ref * ref))
    return float(np.sqrt(np.sum((y - ref) ** 2)) / den) if den > 0 else float(np.sqrt(np.sum(y * y)))
