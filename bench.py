#!/usr/bin/env python3
"""bench.py -- throughput of the libdsp hot path on MI355X, one JSON line on stdout.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload fir|resample|decimate]

Default line (what the driver records): BASELINE.json configs[1] -- a 256-tap FIR over a
2^28-sample complex-float32 stream, device-resident in and out, one GPU.  A "step" is one pass
of the hot path over that batch: one sfe_dsp_fir_process_stream call (the FFT overlap-save
kernel, which also writes the 2 KiB history carry-over).

N > 1: BASELINE.json configs[4] as SURVEY.md 8(d) cfg5 words it -- a FIXED job of 64 independent
channels x 2^24 samples, block-partitioned 64/N channels per rank (shard.channel_block), one rank
per GPU, all of a rank's channels in ONE launch, no data-path collective ("scaling": "strong";
`value` = 2^30 samples / max-over-ranks step time; `roofline.frac` against N x 8 TB/s).  The only
cross-rank traffic is the barrier, the MAX of the elapsed time and of the worst parity figure, and
the SUM of the per-rank output checksums {samples, sum re, sum im, sum |y|^2} (RCCL; SURVEY.md
8(e)); every local channel of every rank is checked.  The round-2 weak-scaling shape (8 channels x
2^25 per rank) is timed as an `other_configs` row.  After the timed region the ranks also cut ONE
stream into spans, send the 256-sample span tails to their right neighbours point to point and
check the seams (`split_stream`).  --channels / --log2n describe the whole
job (per-channel length = 2^log2n / channels).  Started under torch.distributed.run the ranks are taken from the
environment; started bare (`python bench.py --gpus 4`) this script launches the N ranks itself
BEFORE anything touches the GPU and relays rank 0's line.  It never prints an n_gpus=1 line for
--gpus N > 1: a world size that does not match --gpus is an error.

`roofline`      algorithmic bytes of the dominant kernel per launch / its mean duration, timed
                with HIP events on the launch stream inside the timed region (min / max / std /
                median of the K steps beside the mean); `traffic` is the PMC-measured HBM bytes
                per launch from profiles/ -- null if no pass was recorded, and null with
                `traffic_stale: true` if simplefe_amd/csrc/ has changed since that pass.
buffers         plain allocations (torch.empty), whatever the driver hands out: round 4's screening of output candidates and
                its library-built pairs selected on the measured quantity and are gone (the pair allocator
                now lives in the diagnostic library only).  N = 1 headline: three plain pairs, the leg runs on the MEDIAN one
                (`config.pairs`; --plain-pair: the first).  Every float32 input is verified bit for bit against the host
                generator BEFORE the timed steps and again after them; a leg whose input fails is an error row, never timed.
`other_configs` (N = 1, default workload) the other BASELINE.json configs at G = 1, each a short
                timed leg of its own with parity: resample 5/3 (configs[2], in both readings of
                "127-tap polyphase arm": the 381-tap prototype and the 127-tap prototype), decimate by 8
                (configs[3]), the 64-channel FIR on one GPU (configs[4] at G = 1; shared filter, and a
                different filter per channel), the
                complex-tap FIR, and two rows that are not BASELINE configs: the general
                (non-integer-step) rate 1.77 and interpolation by 2.  They run BEFORE the headline's warm-up; their
                parity checks run after all timing.
`precondition_s` seconds of untimed launches of whichever leg runs first, before any warm-up or timed
                step: the chip's first ~100 ms of work after idling run 5-6 % slow (DESIGN.md
                section 6), and W = 5 steps of a 0.8 ms kernel end well inside that.
`cpu_baseline`  the CPU oracle (oracle/, a port of the reference algorithm; oracle/_ref for
                resample/decimate = the reference's own code) timed on this host on a bounded
                sample of the same workload, rank 0, N = 1 only: `value` on one thread (the
                reference is single-threaded), `all_cores` on every host core (spans with
                n_taps-1 samples of overlap), `host_cores` stated.  A reported baseline.
                `configs0_cpu_only` (fir workload) = BASELINE configs[0], the CPU-only plumbing case:
                63-tap real filter, 2^20 real float32 samples, fft_len 1024, same port, one thread.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TOL = 1e-5                # BASELINE.json north_star


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # ~0.2 s of GPU time
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="fir", choices=["fir", "fir_ctaps", "resample", "decimate"])
    ap.add_argument("--log2n", type=int, default=None,
                    help="samples in the whole job = 2^log2n (default per workload: 28 at N=1, 30 for the 64-channel job at N>1; "
                         "resample/decimate: per GPU)")
    ap.add_argument("--algo", default="auto", choices=["auto", "fft", "direct"])
    ap.add_argument("--channels", type=int, default=None,
                    help="channels in the whole job, block-partitioned over the ranks (default 1 at N=1, 64 at N>1: configs[4])")
    ap.add_argument("--input", default="f32", choices=["f32", "u8"],
                    help="u8: the stream is the device wire format, converted on load (fused RX converter, N2)")
    ap.add_argument("--output", default="f32", choices=["f32", "tx10"],
                    help="tx10: the FIR writes the 10-bit transmit wire format (fused TX converter, N2)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-others", action="store_true", help="skip the other_configs legs")
    ap.add_argument("--telemetry", action="store_true",
                    help="after everything else: an UNTIMED run of ~2 s of the headline's launches with rocm-smi's shader clock and package power read "
                         "three times while they run, reported as roofline.telemetry (round 4: the FIR runs the package at its power cap on data, "
                         "DESIGN.md 9; off by default)")
    ap.add_argument("--plain-pair", action="store_true",
                    help="N = 1: the headline on the first pair of plain allocations (default: the MEDIAN of three plain pairs, all three reported)")
    ap.add_argument("--no-calibrate", action="store_true",
                    help="do not call sfe_dsp_fir_calibrate before the headline leg (run the default variant)")
    ap.add_argument("--single-process", action="store_true",
                    help="one process drives all --gpus devices through sfe_dsp_fir_group_* (no torch.distributed)")
    ap.add_argument("--other-steps", type=int, default=20)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--precondition", type=float, default=0.15, help="seconds of untimed launches before any warm-up")
    return ap.parse_args()


# --------------------------------------------------------------------------- self-launch
def spawn_ranks(args):
    """`python bench.py --gpus N` from a bare shell: start the N ranks as children of a parent
    that never initialises the GPU (no exec after GPU init: the parent only counts devices)."""
    import socket
    import torch      # device_count() does not create a GPU context on this image
    one_dev = bool(os.environ.get("SFE_BENCH_ONE_DEVICE"))
    have = torch.cuda.device_count()
    if have < 1:
        raise SystemExit("bench.py needs a GPU: libsfe_dsp has no CPU fallback")
    if have < args.gpus and not one_dev:
        raise SystemExit(f"--gpus {args.gpus} but this node shows {have} GPU(s) "
                         "(SFE_BENCH_ONE_DEVICE=1 rehearses N ranks on one device over gloo)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if one_dev:
        env.setdefault("SFE_DIST_BACKEND", "gloo")     # RCCL refuses two ranks on one device
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    r = subprocess.run(cmd, env=env)
    raise SystemExit(r.returncode)


# ------------------------------------------------------------------------------ helpers
def pmc_traffic(workload_key):
    """(HBM bytes per launch, stale) from the committed PMC summary for this workload, latest round.
    The summary carries a hash of simplefe_amd/csrc/ as it was when the counters were collected
    (scripts/summarise_profiles.py); if the kernel sources have changed since, the bytes are not
    reported: (None, True).  No summary at all: (None, False)."""
    from simplefe_amd import build as _b
    best = None
    pdir = os.path.join(ROOT, "profiles")
    if not os.path.isdir(pdir):
        return None, False
    for f in sorted(os.listdir(pdir)):
        if f.startswith("pmc_") and f.endswith(".json"):
            try:
                d = json.load(open(os.path.join(pdir, f)))
            except Exception:
                continue
            if d.get("workload") == workload_key and d.get("hbm_bytes_per_launch"):
                best = d
    if best is None:
        return None, False
    if best.get("csrc_sha256") != _b.csrc_hash(best.get("csrc_kind")):      # the files of this workload's kernel (older passes: every kernel source)
        return None, True
    return best["hbm_bytes_per_launch"], False


def host_cores():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def _timed_reps(fn, budget_s, max_reps):
    t0 = time.perf_counter()
    fn()
    dt = time.perf_counter() - t0
    reps = min(max(1, int(budget_s / max(dt, 1e-6))), max_reps)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return reps, time.perf_counter() - t0


def _all_cores(run, n_samples, budget_s):
    """run(C) pushes one n_samples-long cf32 stream (I then Q) through C threads, each with its own
    oracle objects on its own span (oracle/: orc_*_stream_mt, pthreads / std::thread)."""
    C = host_cores()
    run(C)                                             # warm
    reps, dt = _timed_reps(lambda: run(C), budget_s, 256)
    return {"value": reps * n_samples / dt / 1e6, "unit": "MS/s", "cores": C,
            "sample": f"{reps} x 2^{n_samples.bit_length() - 1} cf32 samples cut into {C} spans with lead-in, "
                      f"one object per span and thread, {dt:.1f} s"}


def cpu_baseline_fir(taps, budget_s):
    """Oracle port of blkconv (fft_len 4096, blk 3841), I and Q as two real passes."""
    from oracle import binding as orc
    from simplefe_amd import synth
    n = 1 << 20
    hl = len(taps) - 1
    x = synth.synth_cf32(n + hl)
    xr, xi = np.ascontiguousarray(x[0::2]), np.ascontiguousarray(x[1::2])
    cr, ci = orc.Blkconv(taps, 4096), orc.Blkconv(taps, 4096)
    reps, dt = _timed_reps(lambda: (cr.stream(xr[hl:]), ci.stream(xi[hl:])), 0.6 * budget_s, 1024)
    out = {"value": reps * n / dt / 1e6, "unit": "MS/s", "cores": 1, "kind": "port",
           "sample": f"{reps} x 2^20 cf32 samples, blkconv port fft_len 4096 (own float32 FFT; "
                     f"FFTW absent), I and Q as two real passes, {dt:.1f} s"}

    # long enough that every core's span is many blocks and thread start-up does not show: 2^19
    # samples per core (a 256-core host: 2^27); the I and the Q pass run over the same real buffer
    nb = 1 << min(27, max(23, (host_cores() - 1).bit_length() + 19))
    br = np.tile(np.ascontiguousarray(synth.synth_cf32(1 << 21)[0::2]), nb >> 21)
    out["all_cores"] = _all_cores(lambda C: (orc.blkconv_stream_mt(taps, 4096, br, C, want_output=False),
                                             orc.blkconv_stream_mt(taps, 4096, br, C, want_output=False)),
                                  nb, 0.4 * budget_s)
    out["host_cores"] = host_cores()
    # BASELINE configs[0] ("plumbing, CPU only -- report CPU MS/s, no GPU number"): the 63-tap real
    # filter at the block size libdsp/test uses it with, 2^20 real float32 samples through the same port, one thread
    t1 = synth.taps_cfg1()
    x1 = synth.synth_f32(1 << 20)
    c1 = orc.Blkconv(t1, 1024)
    reps, dt = _timed_reps(lambda: c1.stream(x1), 0.5, 64)
    out["configs0_cpu_only"] = {"value": reps * (1 << 20) / dt / 1e6, "unit": "MS/s (real samples)", "cores": 1,
                                "sample": f"{reps} x 2^20 real float32 samples, 63 taps, fft_len 1024 (blk 962), {dt:.1f} s"}
    return out


def cpu_baseline_rs(which, taps, U, rate, budget_s):
    """The reference's own resample/decimate class (oracle/_ref) when present, else the port."""
    from oracle import binding as orc
    from simplefe_amd import synth
    use_ref = orc.ref_lib() is not None
    cls = {("resample", True): orc.RefResample, ("resample", False): orc.Resample,
           ("decimate", True): orc.RefDecimate, ("decimate", False): orc.Decimate}[(which, use_ref)]
    n, B = 1 << 18, 4096
    x = synth.synth_cf32(n)
    xr, xi = np.ascontiguousarray(x[0::2]), np.ascontiguousarray(x[1::2])
    a, b = cls(taps, U, B), cls(taps, U, B)
    reps, dt = _timed_reps(lambda: (a.stream(xr, rate), b.stream(xi, rate)), 0.6 * budget_s, 4096)
    out = {"value": reps * n / dt / 1e6, "unit": "MS/s", "cores": 1, "kind": "reference" if use_ref else "port",
           "sample": f"{reps} x 2^18 cf32 samples, libdsp {which} class chunked at {B}, "
                     f"I and Q as two real passes, {dt:.1f} s"}

    S = int(round(rate * U))
    quantum = S // int(np.gcd(S, U))
    nb = 1 << min(26, max(21, (host_cores() - 1).bit_length() + 17))
    br = np.tile(np.ascontiguousarray(synth.synth_cf32(1 << 20)[0::2]), nb >> 20)
    out["all_cores"] = _all_cores(lambda C: (orc.rs_stream_mt(which, taps, U, B, rate, br, quantum, C, reference=use_ref),
                                             orc.rs_stream_mt(which, taps, U, B, rate, br, quantum, C, reference=use_ref)),
                                  nb, 0.4 * budget_s)
    out["host_cores"] = host_cores()
    return out


# --------------------------------------------------------------------------------- legs
class Leg:
    """One workload on this rank's GPU: buffers, the handle, step(), and the parity check."""


def _p2(v):
    """'2^k' for a power of two, the plain number otherwise."""
    v = int(v)
    return "2^%d" % (v.bit_length() - 1) if v > 0 and v & (v - 1) == 0 else str(v)


def telemetry(ctx, leg, kern_ms):
    """--telemetry: ~2 s of the leg's launches queued (untimed), rocm-smi read three times while they run.  Best effort: any
    failure is reported in the field, never raised (scripts/probes/fir_power.py is the stand-alone form)."""
    import re
    import subprocess
    torch = ctx["torch"]
    try:
        count = max(50, min(4000, int(2000.0 / max(kern_ms, 0.05))))
        for _ in range(count):
            leg.step()
        reads = []
        for _ in range(3):
            time.sleep(0.25)
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=6).stdout
            sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)
            watt = re.search(r"Package Power \(W\): ([0-9.]+)", txt)
            reads.append({"sclk_mhz": int(sclk.group(1)) if sclk else None, "package_w": float(watt.group(1)) if watt else None})
        torch.cuda.synchronize()
        return {"launches_queued": count, "reads_while_running": reads, "source": "rocm-smi --showclocks --showpower"}
    except Exception as e:                      # noqa: BLE001
        try:
            torch.cuda.synchronize()
        except Exception:                       # noqa: BLE001
            pass
        return {"error": "%s: %s" % (type(e).__name__, e)}


def verify_input(ctx, leg, windows=24, W=4096):
    """Is the input the timed launches read still the synthetic stream it was filled with?  (The parity windows feed the
    oracle what the device holds, so an input that lost its data -- zeros -- would pass them, and a FIR on zeros runs 10 %
    faster: DESIGN.md 4.2.)  Windows spread over the first and the last region of leg.input_spec = [(offset, n_floats, channel
    seed)], compared bit for bit with the host generator.  None = nothing to check (derived inputs: u8 legs, shared buffers)."""
    spec, synth = getattr(leg, "input_spec", None), ctx["synth"]
    if not spec:
        return None
    for off, nf, ch in (spec[0], spec[-1]):
        for k in range(windows):
            pos = (nf - W) * k // (windows - 1)
            pos -= pos % 2
            got = leg.x[off + pos: off + pos + W].cpu().numpy()
            if not np.array_equal(got, synth.synth_f32(W, synth.SEED, ch, first=pos)):
                return False
    return True


def make_fir_leg(ctx, name, taps, n, nch, ch0=0, algo="auto", in_fmt="f32", out_fmt="f32", x_share=None, per_channel=False,
                 y_share=None, calibrate=False, median_of_pairs=False):
    """n samples per channel, nch channels on THIS rank, the first of them global channel ch0 (its
    seed): what one rank of a channel-sharded job holds (shard.channel_block)."""
    torch, api, lib, synth, shard = ctx["torch"], ctx["api"], ctx["lib"], ctx["synth"], ctx["shard"]
    dev, stream, L = ctx["dev"], ctx["stream"], ctx["L"]
    leg = Leg()
    leg.name, leg.kind = name, "fir"
    n_gpu = n * nch
    leg.n, leg.nch, leg.n_gpu = n, nch, n_gpu
    leg.seeds = [ch0 + c for c in range(nch)]                  # global channel id = its seed
    if x_share is not None:
        x = x_share
    else:
        x = torch.empty(nch * n * 2, dtype=torch.float32, device=dev)
        for c in range(nch):
            api.check(L.sfe_dsp_synth_fill(x.data_ptr() + c * n * 8, 2 * n, synth.SEED, leg.seeds[c], 0, stream))
    in_bytes, out_bytes = 8.0, 8.0
    src = x
    ctaps = bool(np.iscomplexobj(taps))
    n_taps = taps.shape[-1]
    leg.workload = "%d-tap %sFIR (blkconv law), %s cf32 samples per GPU, %d channel(s) x %s, device-resident in/out" % (
        n_taps, "complex-tap " if ctaps else "", _p2(n_gpu), nch, _p2(n))
    if per_channel:
        leg.workload += ", a different filter per channel"
    leg.key = "fir256_cf32_2p%d%s%s" % (max(n_gpu, 1).bit_length() - 1, "_ctaps" if ctaps else "", "_direct" if algo == "direct" else "")
    if nch > 1:
        leg.key += "_%dch" % nch
    if in_fmt == "u8":
        # wire format: (I,Q) byte pairs.  Derived from the float stream so the parity leg still has a
        # float twin: b = round(127 x) + 128, i.e. x_u8 = (b - 128)/127.
        xb = (torch.round(x * 127.0) + 128.0).clamp_(0, 255).to(torch.uint8)
        x = (xb.to(torch.float32) - 128.0) * (1.0 / 127.0)          # the float twin (exact)
        src, in_bytes = xb, 2.0
        leg.key += "_u8"
        leg.workload += ", u8 (I,Q) wire-format input converted on load"
    leg.x = x
    leg.input_spec = [(c * 2 * n, 2 * n, leg.seeds[c]) for c in range(nch)] if (in_fmt == "f32" and x_share is None) else None
    if per_channel:
        leg.obj = api.Fir(taps, per_channel=True, device=ctx["local_rank"])
        leg.key += "_pctaps"
    else:
        leg.obj = api.Fir(taps, data_complex=True, n_channels=nch, device=ctx["local_rank"],
                          algo={"auto": lib.FIR_ALGO_AUTO, "fft": lib.FIR_ALGO_FFT, "direct": lib.FIR_ALGO_DIRECT}[algo])
    if in_fmt == "u8":
        leg.obj.set_input_format(lib.FMT_U8)
    if out_fmt == "tx10":
        leg.obj.set_output_format(lib.FMT_TX10)
        out_bytes = 2.5
        leg.key += "_tx10"
        leg.workload += ", 10-bit packed transmit wire format out"
        leg.y = torch.empty(nch * (n * 2 // 4) * 5 + 64, dtype=torch.uint8, device=dev)
    else:
        leg.y = y_share if y_share is not None else torch.empty(nch * n * 2, dtype=torch.float32, device=dev)
    leg.out_fmt = out_fmt
    if median_of_pairs and x_share is None and y_share is None and in_fmt == "f32" and out_fmt == "f32":
        # The headline leg only: what a pair of plain allocations gives a read + write stream differs from
        # pair to pair by up to ~8 % on this platform (profiles/r04/NOTES.md).  Three plain pairs, each timed with five launches
        # of the leg's own call after a warm-up, and the leg runs on the MEDIAN one -- not the fastest: a robust draw, not a
        # selection of the best.  All three figures go into the line; outside the timed region.
        pairs = [(x, leg.y)]
        try:
            for _ in range(2):
                xa = torch.empty_like(x)
                for c in range(nch):
                    api.check(L.sfe_dsp_synth_fill(xa.data_ptr() + c * n * 8, 2 * n, synth.SEED, leg.seeds[c], 0, stream))
                pairs.append((xa, torch.empty_like(leg.y)))
        except RuntimeError:
            pass
        if len(pairs) == 3:
            tm = api.Timer()
            for _ in range(40):                  # the chip through its start-up transient before anything is compared
                leg.obj.process_stream(pairs[0][0].data_ptr(), pairs[0][1].data_ptr(), n, stream=stream)
            times = []
            for xs, ys in pairs:
                for _ in range(2):
                    leg.obj.process_stream(xs.data_ptr(), ys.data_ptr(), n, stream=stream)
                tm.start(stream)
                for _ in range(5):
                    leg.obj.process_stream(xs.data_ptr(), ys.data_ptr(), n, stream=stream)
                tm.stop(stream)
                times.append(tm.elapsed_ms() / 5)
            mid = int(np.argsort(times)[1])
            x = src = leg.x = pairs[mid][0]
            leg.y = pairs[mid][1]
            leg.pairs = {"pairs_timed_ms": [round(v, 4) for v in times], "kept": mid, "rule": "the MEDIAN of three plain pairs, by five launches each"}
            torch.cuda.synchronize()
            leg.obj.reset()
        del pairs
        torch.cuda.empty_cache()
    leg.bytes_per_launch = (in_bytes + out_bytes) * n_gpu       # 8 B read + 8 B written per sample
    leg.kernel = "fir_fft4096_kernel" if algo != "direct" else "poly_tiled_kernel"
    leg.taps = taps
    leg.n_out = n
    sp, yp = src.data_ptr(), leg.y.data_ptr()
    leg._src = src

    def step():
        leg.obj.process_stream(sp, yp, n, stream=stream)
    leg.step = step
    # Stream calls never measure (api_fir.hip: fir_pick_variant is a table look-up; register loads by default).
    # The headline leg asks for the measurement explicitly, untimed, before any warm-up:
    # sfe_dsp_fir_calibrate on rank 0, the choice handed to every rank so that all of them run ONE kernel
    # (independently calibrating ranks were noise in a strong-scaling curve).
    step()
    torch.cuda.synchronize()
    leg.variant = None
    if calibrate:
        chosen, cal, ms = 0, 0, []
        if ctx["rank"] == 0:
            chosen = leg.obj.calibrate(sp, yp, n, stream=stream)
            _, cal, ms = leg.obj.get_variant()
        if ctx["world"] > 1:
            chosen = int(round(shard.sum_over_ranks([float(chosen)], ctx["red_dev"])[0]))     # only rank 0 contributes
        leg.obj.set_variant(chosen)
        leg.variant = {"ran": lib.FIR_VARIANT_NAMES.get(chosen, str(chosen)),
                       "chosen_by": "sfe_dsp_fir_calibrate on rank 0, untimed" + (", broadcast to all ranks" if ctx["world"] > 1 else "")}
        if cal:
            leg.variant["median_ms"] = {lib.FIR_VARIANT_NAMES[i]: round(m, 4) for i, m in enumerate(ms) if m > 0}
    else:
        step()
        v, _, _ = leg.obj.get_variant()
        leg.variant = {"ran": lib.FIR_VARIANT_NAMES.get(v, str(v)), "chosen_by": "default (nothing measured)"}

    def check(full):
        """Windows of the LAST step's output against the oracle (history = the tail of the same
        buffer fed in the previous step).  Returns the worst rel-RMS (tx10: fraction of bytes off)."""
        from oracle import binding as orc
        W, hlen = 1 << 13, n_taps - 1
        chans = sorted(set([0, nch - 1] if not full else range(nch)))
        starts = [0, 3840 - 100, n // 2 - 77, n - W]
        if nch > 2:
            starts = [0, n - W]
        worst, count = 0.0, 0
        blk = lambda t, v: orc.Blkconv(t, 4096).stream(np.ascontiguousarray(v))
        for c in chans:
            tc = taps[c] if per_channel else taps
            tr = np.ascontiguousarray(np.real(tc), dtype=np.float32)
            ti = np.ascontiguousarray(np.imag(tc), dtype=np.float32) if ctaps else None
            xc = leg.x[2 * n * c: 2 * n * (c + 1)]
            for s0 in starts:
                s0 = max(0, min(s0, n - W))
                s0 -= s0 % 2                               # whole 5-byte groups in the tx10 form
                lo = s0 - hlen
                if lo >= 0:
                    seg = xc[2 * lo: 2 * (s0 + W)].cpu().numpy()
                else:
                    seg = np.concatenate([xc[2 * (n + lo):].cpu().numpy(), xc[: 2 * (s0 + W)].cpu().numpy()])
                xr, xi = seg[0::2], seg[1::2]
                if ctaps:
                    ref_r, ref_i = (blk(tr, xr) - blk(ti, xi))[hlen:], (blk(tr, xi) + blk(ti, xr))[hlen:]
                else:
                    ref_r, ref_i = blk(tr, xr)[hlen:], blk(tr, xi)[hlen:]
                if out_fmt == "tx10":
                    # packed 10-bit output: the oracle's floats are packed by the oracle's converter and the
                    # bytes compared; the figure is the fraction of bytes that differ (a sample that sits
                    # on a quantiser step can land one code apart: the two float32 FFTs round differently)
                    ref = np.empty(2 * W, np.float32)
                    ref[0::2], ref[1::2] = ref_r, ref_i
                    want = orc.tx_f32_to_10bit(ref)
                    g0 = (n * 2 // 4) * 5 * c + (2 * s0 // 4) * 5
                    got = leg.y[g0: g0 + len(want)].cpu().numpy()
                    worst = max(worst, float(np.mean(got != want)))
                else:
                    got = leg.y[2 * (n * c + s0): 2 * (n * c + s0 + W)].cpu().numpy()
                    worst = max(worst, synth.rel_rms(got[0::2], ref_r), synth.rel_rms(got[1::2], ref_i))
                count += 1
        return worst, count, W
    leg.check = check
    return leg


def make_rs_leg(ctx, which, log2n, in_fmt="f32", short_proto=False):
    torch, api, lib, synth = ctx["torch"], ctx["api"], ctx["lib"], ctx["synth"]
    dev, stream, L = ctx["dev"], ctx["stream"], ctx["L"]
    leg = Leg()
    leg.name, leg.kind = which, "rs"
    n = 1 << log2n
    leg.n, leg.nch, leg.n_gpu = n, 1, n
    if which == "resample" and short_proto:
        taps, U, S = synth.taps_cfg3_short(), 3, 5
        leg.workload = "rational resample 5/3, 127-tap prototype (43 per arm), 2^%d cf32 in" % log2n
        leg.key = "resample5o3_127_cf32_2p%d" % log2n
    elif which == "resample":
        taps, U, S = synth.taps_cfg3(), 3, 5
        leg.workload = "rational resample 5/3, 381-tap prototype (127 per arm), 2^%d cf32 in" % log2n
        leg.key = "resample5o3_cf32_2p%d" % log2n
    else:
        taps, U, S = synth.taps_cfg4(), 1, 8
        leg.workload = "decimate by 8, 64-tap anti-alias FIR, 2^%d cf32 in" % log2n
        leg.key = "decimate8_cf32_2p%d" % log2n
    rate = float(np.float32(S) / np.float32(U))
    leg.taps, leg.U, leg.S, leg.rate = taps, U, S, rate
    x = torch.empty(n * 2, dtype=torch.float32, device=dev)
    api.check(L.sfe_dsp_synth_fill(x.data_ptr(), 2 * n, synth.SEED, 0, 0, stream))
    src, in_bytes = x, 8.0
    if in_fmt == "u8":
        xb = (torch.round(x * 127.0) + 128.0).clamp_(0, 255).to(torch.uint8)
        x = (xb.to(torch.float32) - 128.0) * (1.0 / 127.0)
        src, in_bytes = xb, 2.0
        leg.key += "_u8"
        leg.workload += ", u8 (I,Q) wire-format input converted on load"
    leg.x, leg._src = x, src
    out_cap = n * U // S + 8
    leg.y = torch.empty(out_cap * 2, dtype=torch.float32, device=dev)
    leg.input_spec = [(0, 2 * n, 0)] if in_fmt != "u8" else None
    leg.obj = api.Rs(taps, U, 4096, mode=lib.RS_RESAMPLE if which == "resample" else lib.RS_DECIMATE,
                     data_complex=True, n_channels=1, device=ctx["local_rank"])
    if in_fmt == "u8":
        leg.obj.set_input_format(lib.FMT_U8)
    # the library picks the transform-domain kernel for long filters on cf32 streams (api_plans.hip: get_fft_plan)
    leg.kernel = "poly_fft256_kernel" if which == "resample" and not short_proto else "poly_tiled_kernel"
    leg.k_before = 0          # outputs produced by all calls before the most recent one
    leg.n_out = 0
    leg.calls = 0
    sp, yp = src.data_ptr(), leg.y.data_ptr()

    def step():
        leg.k_before += leg.n_out
        leg.n_out = leg.obj.process_stream(sp, n, yp, out_cap, rate, stream=stream)
        leg.calls += 1
    leg.step = step
    step()                    # sizes the launch: outputs per call differ by at most one between calls
    leg.bytes_per_launch = in_bytes * n + 8.0 * leg.n_out

    def check(full):
        """Windows of the LAST call's output.  The handle has seen the same buffer `calls` times, so
        the last call continues a virtual stream x|x|x|...: its local output i is global output
        K = k_before + i at upsampled position K*S (resample.cxx:125-150: the carried pos / leftover
        state is exactly this bookkeeping), and the oracle is restarted a0 samples into that virtual
        stream with enough history in front (finite memory)."""
        from oracle import binding as orc
        g = int(np.gcd(S, U))
        per = S // g
        plen = (len(taps) + U - 1) // U + 1
        W = 1 << 13
        worst, count = 0.0, 0
        base_in = (leg.calls - 1) * n                 # global input index of this call's sample 0
        for i0 in (0, leg.n_out // 2, leg.n_out - W - 8):
            K0 = leg.k_before + i0
            nin0 = (K0 * S) // U
            a0 = max(0, ((nin0 - plen - per) // per) * per)
            j0 = K0 - a0 * U // S
            span_end = min(((K0 + W) * S) // U + 2, base_in + n)
            idx0, idx1 = a0 - base_in, span_end - base_in      # relative to this call's buffer
            if idx0 >= 0:
                seg = leg.x[2 * idx0: 2 * idx1].cpu().numpy()
            else:                                          # reaches back into the previous call = the buffer's tail
                seg = np.concatenate([leg.x[2 * (n + idx0):].cpu().numpy(), leg.x[: 2 * idx1].cpu().numpy()])
            got = leg.y[2 * i0: 2 * (i0 + W)].cpu().numpy()
            for part in (0, 1):
                ref, _ = orc.Decimate(taps, U, 4096).stream(np.ascontiguousarray(seg[part::2]), rate)
                m = min(W, len(ref) - j0)
                worst = max(worst, synth.rel_rms(got[part::2][:m], ref[j0:j0 + m]))
            count += 1
        return worst, count, W
    leg.check = check
    return leg


def make_general_rate_leg(ctx, log2n=28, rate=1.77, interpolate=0, decimate=0, real=False, u8=False):
    """the general (non-integer-step) rate at bulk size -- BASELINE cfg3's 381-tap prototype in 3 phases at
    rate 1.77 (the rate of the reference's own driver, libdsp/test/test_decimate.py:24), 2^28 cf32 in.  The library's default
    dispatch takes the transform-domain kernel (poly_gen.hip).  Not a BASELINE config: an other_configs row.
    Parity: the time law makes windows of a later call awkward to restart on the CPU, so the check runs the SAME handle
    type from a fresh state over the stream's first 2^20 samples and compares ALL of that call's outputs (and their count)
    with the oracle fed the same samples."""
    torch, api, lib, synth = ctx["torch"], ctx["api"], ctx["lib"], ctx["synth"]
    dev, stream, L = ctx["dev"], ctx["stream"], ctx["L"]
    leg = Leg()
    leg.name, leg.kind = "general_rate", "rs"
    n = 1 << log2n
    taps, U = synth.taps_cfg3(), 3
    rate = float(np.float32(rate))
    leg.n, leg.nch, leg.n_gpu = n, 1, n
    leg.workload = "general-rate resample, rate %.2f (non-integer step %.2f), 381-tap prototype in 3 phases, 2^%d cf32 in" % (rate, rate * U, log2n)
    leg.key = "resample_rate1p77_cf32_2p%d" % log2n
    if interpolate:
        # the one thing `resample` does and `decimate` refuses (libdsp/resample.cxx:91 against decimate.cxx:75-78): MORE
        # outputs than inputs.  Interpolation by `interpolate`: rate 1 / U, 32 taps per polyphase arm.
        U = int(interpolate)
        taps = synth.lowpass_taps(32 * U, 0.9 / U, gain=float(U))
        rate = float(np.float32(1.0) / np.float32(U))
        leg.name = "interpolate_x%d" % U
        leg.workload = "interpolate x%d (resample at rate 1/%d), %d-tap prototype (32 per arm), 2^%d cf32 in" % (U, U, len(taps), log2n)
        leg.key = "resample_x%d_cf32_2p%d" % (U, log2n)
    if decimate:
        # an integer step with no compile-time kernel (round 5): the runtime-shape kernel whose tile arrives by LDS-DMA and is read in place
        # (poly_rt_dma.hip) -- decimate by `decimate`, 32 taps
        U = 1
        taps = synth.lowpass_taps(32, 0.9 / decimate, gain=1.0)
        rate = float(decimate)
        leg.name = "decimate_by_%d" % decimate
        leg.workload = "decimate by %d, 32-tap anti-alias FIR (a shape outside the compiled tables), 2^%d cf32 in" % (decimate, log2n)
        leg.key = "decimate%d_cf32_2p%d" % (decimate, log2n)
    # the same 2^(log2n + 3) bytes of synthetic float32 stream serve every form of the input: complex samples, (real) twice as many real
    # samples -- libdsp's native type, the reference's classes take float* --, or (u8) its first 2^(log2n + 1) bytes read as the receive
    # wire format, u8 offset-binary (I, Q) pairs (gr-simplefe/lib/source_c_impl.cc)
    w = 1 if real else 2
    if real:
        n = 2 * n
        leg.n = leg.n_gpu = n
        leg.name += "_real"
        leg.workload = "decimate by %d, 32-tap anti-alias FIR, REAL float32 stream (libdsp's native type), 2^%d real samples in" % (decimate, log2n + 1)
        leg.key = "decimate%d_f32_2p%d" % (decimate, log2n + 1)
    if u8:
        leg.name += "_u8"
        leg.workload = "decimate by %d, 32-tap anti-alias FIR, u8 wire-format input (I, Q byte pairs) -> cf32, 2^%d complex samples in" % (decimate, log2n)
        leg.key = "decimate%d_u8_2p%d" % (decimate, log2n)
    x = torch.empty((1 << log2n) * 2, dtype=torch.float32, device=dev)
    api.check(L.sfe_dsp_synth_fill(x.data_ptr(), x.numel(), synth.SEED, 0, 0, stream))
    out_cap = int(n / rate) + 4096
    leg.x = x
    leg.y = torch.empty(out_cap * w, dtype=torch.float32, device=dev)
    leg.input_spec = [(0, x.numel(), 0)]
    leg.obj = api.Rs(taps, U, 4096, mode=lib.RS_RESAMPLE, data_complex=not real, n_channels=1, device=ctx["local_rank"])
    if u8:
        leg.obj.set_input_format(lib.FMT_U8)
    leg.kernel = "poly_rt1_kernel" if interpolate else (("poly_int4_dma_kernel" if real else "poly_rt_dma_kernel") if decimate else "poly_gen4096_kernel")
    leg.n_out = 0
    sp, yp = x.data_ptr(), leg.y.data_ptr()

    def step():
        leg.n_out = leg.obj.process_stream(sp, n, yp, out_cap, rate, stream=stream)
    leg.step = step
    step()
    leg.bytes_per_launch = (2.0 if u8 else 4.0 * w) * n + 4.0 * w * leg.n_out

    def check(full):
        from oracle import binding as orc
        m = 1 << 20
        r = api.Rs(taps, U, 4096, mode=lib.RS_RESAMPLE, data_complex=not real, n_channels=1, device=ctx["local_rank"])
        if u8:
            r.set_input_format(lib.FMT_U8)
        if not interpolate and not decimate:
            r.set_algo(lib.RS_ALGO_FFT)         # the kernel the timed leg's bulk calls take, also at this size
        y = torch.empty((int(m / rate) + 64) * w, dtype=torch.float32, device=dev)
        k = r.process_stream(sp, m, y.data_ptr(), y.numel() // w, rate, stream=stream)
        torch.cuda.synchronize()
        got = y[: w * k].cpu().numpy()
        if u8:                                  # the oracle's own converter over the same bytes (gr-simplefe/lib/source_c_impl.cc's (b - 128) / 127)
            xin = orc.rx_u8_to_f32(x[: m // 2].cpu().numpy().view(np.uint8))
        else:
            xin = x[: w * m].cpu().numpy()
        worst = 0.0
        for part in range(w):
            ref, _ = orc.Resample(taps, U, 4096).stream(np.ascontiguousarray(xin[part::w]), rate)
            if len(ref) != k:
                return 1.0, 1, k                # a different number of outputs is a failure whatever the values
            worst = max(worst, synth.rel_rms(got[part::w], ref))
        return worst, 1, k
    leg.check = check
    return leg


def group_leg(ctx, taps, total_channels, n_ch, devices, steps, warmup):
    """ONE process, len(devices) devices: the channel partition made inside the library
    (sfe_dsp_fir_group_*, include/sfe_dsp.h) instead of one rank per GPU.  Launches go to every
    device before anything waits; the time is the wall clock around `steps` group calls and one
    sync.  Every shard's first and last channel is checked against the oracle (first and last
    window), and the float64 checksum over all outputs is what the rank-per-GPU form all-reduces."""
    torch, api, synth, L = ctx["torch"], ctx["api"], ctx["synth"], ctx["L"]
    from oracle import binding as orc
    grp = api.FirGroup(taps, total_channels, devices)
    sh = grp.shards()
    xs, ys = [], []
    for d, f, c, _, _ in sh:
        with torch.cuda.device(d):
            x = torch.empty(c * n_ch * 2, dtype=torch.float32, device="cuda:%d" % d)
            for k in range(c):
                api.check(L.sfe_dsp_synth_fill(x.data_ptr() + k * n_ch * 8, 2 * n_ch, synth.SEED, f + k, 0, None))
            torch.cuda.synchronize(d)
            xs.append(x)
            ys.append(torch.empty_like(x))
    pin, pout = [x.data_ptr() for x in xs], [y.data_ptr() for y in ys]
    for _ in range(max(2, warmup)):
        grp.process_stream(pin, pout, n_ch)
    grp.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        grp.process_stream(pin, pout, n_ch)
    grp.sync()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    # parity: history = the tail of the same buffer (every call re-reads the same input)
    W, hlen, worst, count = 1 << 13, len(taps) - 1, 0.0, 0
    blk = lambda v: orc.Blkconv(np.ascontiguousarray(taps, dtype=np.float32), 4096).stream(np.ascontiguousarray(v))
    csum = [0.0, 0.0, 0.0, 0.0]
    for (d, f, c, _, _), x, y in zip(sh, xs, ys):
        for k in sorted({0, c - 1}):
            xc, yc = x[2 * n_ch * k: 2 * n_ch * (k + 1)], y[2 * n_ch * k: 2 * n_ch * (k + 1)]
            for s0 in (0, n_ch - W):
                seg = (np.concatenate([xc[2 * (n_ch - hlen):].cpu().numpy(), xc[: 2 * W].cpu().numpy()]) if s0 == 0
                       else xc[2 * (s0 - hlen): 2 * (s0 + W)].cpu().numpy())
                got = yc[2 * s0: 2 * (s0 + W)].cpu().numpy()
                worst = max(worst, synth.rel_rms(got[0::2], blk(seg[0::2])[hlen:]), synth.rel_rms(got[1::2], blk(seg[1::2])[hlen:]))
                count += 1
        for k in range(c):
            v = y[2 * n_ch * k: 2 * n_ch * (k + 1)]
            for off in range(0, 2 * n_ch, 1 << 27):
                w = v[off: off + (1 << 27)].view(-1, 2).to(torch.float64)
                csum[1] += float(w[:, 0].sum().item())
                csum[2] += float(w[:, 1].sum().item())
                csum[3] += float((w * w).sum().item())
        csum[0] += float(c * n_ch)
    grp.close()
    job = float(total_channels) * n_ch
    return {"ms": ms, "value": job / (ms * 1e-3) / 1e6, "unit": "MS/s",
            "frac": 16.0 * job / (ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * len(set(devices))),
            "devices": list(devices), "shards": [[d, f, c] for d, f, c, _, _ in sh],
            "parity": {"rel_rms_max": worst, "windows": count, "window_len": W, "tol": TOL, "ok": bool(worst <= TOL)},
            "checksum": {"samples": csum[0], "sum_re": csum[1], "sum_im": csum[2], "sum_abs2": csum[3]},
            "timed": "wall clock around %d sfe_dsp_fir_group_process_stream calls + one sfe_dsp_fir_group_sync" % steps}


def main_single_process(args):
    """`python bench.py --gpus N --single-process`: BASELINE configs[4] (64 channels x 2^24) driven from ONE
    process through sfe_dsp_fir_group_* over N devices -- the form a C++ flowgraph would use.  Same
    contract line; SFE_BENCH_ONE_DEVICE=1 rehearses it with N blocks on device 0."""
    import torch
    from simplefe_amd import api, lib, shard, synth
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libsfe_dsp has no CPU fallback")
    one_dev = bool(os.environ.get("SFE_BENCH_ONE_DEVICE"))
    if torch.cuda.device_count() < args.gpus and not one_dev:
        raise SystemExit(f"--gpus {args.gpus} but this node shows {torch.cuda.device_count()} GPU(s)")
    devices = [0] * args.gpus if one_dev else list(range(args.gpus))
    ctx = {"torch": torch, "api": api, "lib": lib, "synth": synth, "shard": shard, "L": lib.load()}
    total_channels = args.channels or 64
    log2n = args.log2n or 30
    n_ch = (1 << log2n) // total_channels
    g = group_leg(ctx, synth.taps_cfg2(), total_channels, n_ch, devices, args.steps, args.warmup)
    out = {"metric": "complex-float32 MS/s through 256-tap blkconv FIR; % of HBM roofline", "value": g["value"], "unit": "MS/s",
           "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": g["ms"], "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "%d independent cf32 channels x %s samples, 256-tap FIR (blkconv law), ONE process driving %d device(s) "
                                  "through sfe_dsp_fir_group_*, device-resident in/out" % (total_channels, _p2(n_ch), args.gpus),
                      "channels_total": total_channels, "samples_per_channel": n_ch, "devices": g["devices"], "shards": g["shards"],
                      "sharding": "contiguous channel blocks per device inside the library, no data-path exchange"},
           "roofline": {"bound": "hbm", "achieved": 16.0 * total_channels * n_ch / (g["ms"] * 1e-3) / 1e9,
                        "peak": HBM_PEAK_GBS * len(set(devices)), "unit": "GB/s", "frac": g["frac"], "traffic": None,
                        "kernel": "fir_fft4096_kernel", "kernel_ms": g["ms"],
                        "note": "kernel_ms = wall time per group call (all devices in flight together); " + g["timed"]},
           "parity": g["parity"], "checksum": g["checksum"]}
    print(json.dumps(out), flush=True)
    if not g["parity"]["ok"]:
        raise SystemExit(3)


def time_leg(ctx, leg, steps, warmup):
    """W untimed steps, then exactly `steps` timed ones between barriers; returns
    (wall seconds max over ranks, the HIP-event ms of every timed step on the launch stream)."""
    api, shard, stream = ctx["api"], ctx["shard"], ctx["stream"]
    for _ in range(warmup):
        leg.step()
    timers = [api.Timer() for _ in range(steps)]
    ctx["barrier"]()
    t0 = time.perf_counter()
    for k in range(steps):
        timers[k].start(stream)
        leg.step()
        timers[k].stop(stream)
    ctx["barrier"]()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0, ctx["red_dev"])
    return elapsed, [t.elapsed_ms() for t in timers]


def precondition(ctx, leg, seconds=0.15):
    """Untimed: run the leg until `seconds` of wall time have passed.  The chip's first ~100 ms of work
    after idling run 5-6 % slower (power controller / clock start-up transient: a 381-tap resample launch
    measures 0.739 ms after 15 warm-up launches and 0.694 ms after 100); five warm-up steps of a 0.8 ms
    kernel end well inside it.  Outside every timed region, before the W warm-up steps; reported in the
    line as `precondition_s`."""
    torch = ctx["torch"]
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            leg.step()
        torch.cuda.synchronize()


def checksum(ctx, leg):
    """{samples, sum re, sum im, sum |y|^2} of this rank's last output, float64 on the device."""
    torch = ctx["torch"]
    if leg.y.dtype != torch.float32:
        return [float(leg.n_out * leg.nch), float(leg.y.sum(dtype=torch.float64).item()), 0.0, 0.0]
    sre = sim = sq = 0.0
    n2 = 2 * leg.n_out
    stride = leg.y.numel() // leg.nch if leg.nch > 1 else leg.y.numel()
    for c in range(leg.nch):
        v = leg.y[c * stride: c * stride + n2]
        for off in range(0, n2, 1 << 27):                 # bounded float64 temporaries
            w = v[off: off + (1 << 27)].view(-1, 2).to(torch.float64)
            sre += float(w[:, 0].sum().item())
            sim += float(w[:, 1].sum().item())
            sq += float((w * w).sum().item())
    return [float(leg.n_out * leg.nch), sre, sim, sq]


def split_stream_check(ctx, taps):
    """N > 1 only, outside every timed region: ONE stream cut into one span per rank (SURVEY.md 8(e)
    row 3).  Every rank sends the last 256 samples of its span to its right neighbour POINT TO POINT
    (shard.halo_from_left: RCCL send/recv, 2 KiB over one xGMI link -- the path's one real exchange
    step), loads what it received as its handle's history and filters its span; the first window of
    every span -- the samples that depend on the halo -- is checked against the oracle on the uncut
    stream.  If the point-to-point exchange raises on any rank, all ranks agree on that (an
    all-reduce) and repeat the exchange as an all-gather of the span tails; which one carried the
    halo is reported.  Never fatal and never one-sided: every collective is reached by every rank
    whatever happened locally, and a failure is reported in the line."""
    import torch.distributed as dist
    torch, api, synth, shard = ctx["torch"], ctx["api"], ctx["synth"], ctx["shard"]
    L, dev, stream, rank, world = ctx["L"], ctx["dev"], ctx["stream"], ctx["rank"], ctx["world"]
    n_total, HL, W = world << 22, 256, 4096
    first, count = shard.span_block(n_total, world, rank, quantum=3840)
    on_gpu = dist.get_backend() == "nccl"
    cdev = dev if on_gpu else torch.device("cpu")
    err, worst, x = None, float("inf"), None
    tail = torch.zeros(2 * HL, dtype=torch.float32, device=cdev)
    try:
        x = torch.empty(2 * count, dtype=torch.float32, device=dev)
        api.check(L.sfe_dsp_synth_fill(x.data_ptr(), 2 * count, synth.SEED, 77, 2 * first, stream))
        torch.cuda.synchronize()
        tail = x[-2 * HL:].to(cdev).contiguous()
    except Exception as e:
        err = "%s: %s" % (type(e).__name__, e)
    halo, p2p_err = None, None
    try:
        halo = shard.halo_from_left(tail, 2 * HL)               # exchange: send right / receive from the left
    except Exception as e:
        p2p_err = "%s: %s" % (type(e).__name__, e)
    any_failed = shard.max_over_ranks(1.0 if p2p_err else 0.0, ctx["red_dev"]) > 0      # every rank, always
    exchange = "point-to-point send/recv of the span tail to the right neighbour (%s)" % dist.get_backend()
    if any_failed:
        tails = torch.zeros(world * 2 * HL, dtype=torch.float32, device=cdev)
        dist.all_gather_into_tensor(tails, tail)               # the fallback: every rank, always
        halo = tails[(rank - 1) * 2 * HL: rank * 2 * HL] if rank > 0 else torch.zeros_like(tail)
        exchange = "FALLBACK all-gather of the span tails (%s); point-to-point failed" % dist.get_backend()
    if on_gpu:
        torch.cuda.synchronize()        # RCCL orders torch's stream only; the handle below runs on the library's
    try:
        if err is None:
            from oracle import binding as orc
            f = api.Fir(taps, data_complex=True, device=ctx["local_rank"])
            if rank > 0:
                h = halo.to(dev).contiguous()
                f.load_history(h.data_ptr(), HL, stream=stream)
            y = torch.empty(2 * count, dtype=torch.float32, device=dev)
            f.process_stream(x.data_ptr(), y.data_ptr(), count, stream=stream)
            torch.cuda.synchronize()
            lo = max(0, first - (len(taps) - 1))
            seg = synth.synth_cf32(first + W - lo, ch=77, first_sample=lo)      # the uncut stream around the cut
            got = y[: 2 * W].cpu().numpy()
            worst = 0.0
            for part in (0, 1):
                ref = orc.Blkconv(taps, 4096).stream(np.ascontiguousarray(seg[part::2]))[first - lo:]
                worst = max(worst, synth.rel_rms(got[part::2], ref))
    except Exception as e:
        err, worst = "%s: %s" % (type(e).__name__, e), float("inf")
    worst = shard.max_over_ranks(worst, ctx["red_dev"])       # every rank, always
    out = {"ok": bool(worst <= TOL), "rel_rms_max": worst if worst != float("inf") else None, "samples": n_total,
           "spans": world, "halo_samples": HL, "exchange": exchange}
    if p2p_err:
        out["p2p_error"] = p2p_err
    if err:
        out["error"] = err
    return out


def main():
    args = parse()
    if args.single_process:
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
            raise SystemExit("--single-process is one process by definition: start it bare, not under torch.distributed.run")
        return main_single_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)                 # does not return
    from simplefe_amd import shard
    rank, local_rank, world = shard.env_ranks()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a different job size")

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libsfe_dsp has no CPU fallback")
    if os.environ.get("SFE_BENCH_ONE_DEVICE"):      # rehearsal: every rank on cuda:0 (with SFE_DIST_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    shard.init_process_group(dev)            # RCCL when WORLD_SIZE > 1
    red_dev = dev if (world > 1 and dist.get_backend() == "nccl") else None

    from simplefe_amd import api, lib, synth
    L = lib.load()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ctx = {"torch": torch, "api": api, "lib": lib, "synth": synth, "shard": shard, "L": L, "dev": dev,
           "red_dev": red_dev, "stream": torch.cuda.current_stream().cuda_stream, "rank": rank,
           "local_rank": local_rank, "world": world, "barrier": barrier}
    ctx_red = red_dev       # where the control-plane reductions live: the GPU under RCCL, the host under gloo

    wl = args.workload
    total_channels = 1
    if wl in ("fir", "fir_ctaps"):
        # --channels / --log2n describe the WHOLE JOB.  N = 1: configs[1], one channel of 2^28.  N > 1:
        # configs[4] as SURVEY 8(d) cfg5 words it -- 64 channels x 2^24, block-partitioned 64/N per rank
        # (a fixed job: strong scaling); rank r holds channels channel_block(64, N, r), seed = global id.
        total_channels = args.channels or (1 if world == 1 else 64)
        log2n = args.log2n or (28 if world == 1 else 30)
        if (1 << log2n) % total_channels:
            raise SystemExit("--channels must divide 2^log2n")
        n_ch = (1 << log2n) // total_channels
        ch0, nch = shard.channel_block(total_channels, world, rank)
        if nch < 1:
            raise SystemExit(f"--gpus {world} but only {total_channels} channel(s): a rank would hold none")
        taps = synth.taps_cfg2()
        if wl == "fir_ctaps":
            tr, ti = synth.complex_taps(256, 0.2)
            taps = (tr + 1j * ti).astype(np.complex64)
        head = make_fir_leg(ctx, wl, taps, n_ch, nch, ch0=ch0, algo=args.algo, in_fmt=args.input, out_fmt=args.output,
                            calibrate=not args.no_calibrate, median_of_pairs=(world == 1 and not args.plain_pair))
    else:
        head = make_rs_leg(ctx, wl, args.log2n or (28 if wl == "resample" else 30), in_fmt=args.input)

    # ---- the other BASELINE configs at G = 1 (short legs; built and timed before the headline)
    others, other_errors = [], []
    plain = wl == "fir" and not args.no_others and args.input == "f32" and args.output == "f32" \
        and args.algo == "auto" and not args.log2n and not args.channels
    if world == 1 and plain:
        tr, ti = synth.complex_taps(256, 0.2)
        makers = [lambda: make_rs_leg(ctx, "resample", 28), lambda: make_rs_leg(ctx, "resample", 28, short_proto=True),
                  lambda: make_rs_leg(ctx, "decimate", 30),
                  lambda: make_general_rate_leg(ctx),
                  lambda: make_general_rate_leg(ctx, interpolate=2),
                  lambda: make_general_rate_leg(ctx, decimate=7),
                  lambda: make_general_rate_leg(ctx, decimate=3, real=True),
                  lambda: make_general_rate_leg(ctx, decimate=7, u8=True),
                  lambda: make_fir_leg(ctx, "fir_64ch", synth.taps_cfg2(), 1 << 24, 64),
                  lambda: make_fir_leg(ctx, "fir_64ch_pctaps", synth.taps_per_channel(64), 1 << 24, 64, per_channel=True,
                                       x_share=next(l.x for l in others if l.name == "fir_64ch")),
                  lambda: make_fir_leg(ctx, "fir_ctaps", (tr + 1j * ti).astype(np.complex64), 1 << 28, 1, x_share=head.x)]
    elif plain:
        # N > 1: the round-2 weak-scaling shape beside the fixed 64-channel job -- 8 channels x 2^25 per
        # rank whatever N is (the same per-GPU launch at every N; at N = 8 it is 64 channels of 2^25)
        makers = [lambda: make_fir_leg(ctx, "fir_weak", synth.taps_cfg2(), 1 << 25, 8, ch0=8 * rank)]
    else:
        makers = []
    for mk in makers:      # an extra leg that cannot be set up is reported, never allowed to take the headline with it
        try:
            others.append(mk())
        except Exception as e:
            other_errors.append({"error": "%s: %s" % (type(e).__name__, e)})
    if world > 1 and makers:
        # the extra legs time and reduce collectively: they run only if EVERY rank could set all of them up
        failed_somewhere = shard.max_over_ranks(1.0 if other_errors else 0.0, ctx_red) > 0
        if failed_somewhere:
            others = []
            other_errors = other_errors or [{"error": "an other_configs leg could not be set up on another rank"}]
    other_rows = []
    def input_held(leg):
        """verify_input agreed over the ranks (every rank reaches the reduction): None = nothing to check"""
        held = verify_input(ctx, leg)
        bad = shard.max_over_ranks(1.0 if held is False else 0.0, ctx_red) > 0
        return None if (held is None and not bad) else (not bad)

    # every input verified BEFORE anything is timed, all of them in one go: the read-backs leave the GPU idle, and the chip's first
    # ~100 ms after idling run 5-6 % slow -- a check between two legs would put the next leg's timed steps into that transient
    # (round 5's first version did exactly that: headline 0.810 ms against 0.779 for the same pair a few seconds earlier)
    timed_others = []
    for leg in others:
        if input_held(leg) is False:         # never timed: an input that lost its data runs the kernels faster
            other_errors.append({"workload": leg.workload, "error": "the input is not the synthetic stream BEFORE the timed steps; leg not timed"})
        else:
            timed_others.append(leg)
    if input_held(head) is False:
        raise SystemExit("bench.py: the headline input is not the synthetic stream before the timed steps; nothing was timed")
    precondition(ctx, timed_others[0] if timed_others else head, args.precondition)      # whichever leg runs first takes the chip through its start-up transient
    for leg in timed_others:
        el, kms = time_leg(ctx, leg, args.other_steps, 3)
        kmean = shard.max_over_ranks(float(np.mean(kms)), ctx_red)
        row = {"workload": leg.workload, "steps": args.other_steps, "ms": kmean,
               "value": world * leg.n_gpu / (el / args.other_steps) / 1e6, "unit": "MS/s (input)",
               "kernel": leg.kernel, "algorithmic_bytes_per_launch": leg.bytes_per_launch,
               "frac": leg.bytes_per_launch / (kmean * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if getattr(leg, "variant", None):
            row["variant"] = leg.variant["ran"]
        if leg.name == "decimate":
            # this launch runs in one of two modes ~6 % apart that belong to how the process's 8 GiB + 1 GiB buffers
            # happen to be backed physically, not to the kernel (profiles/r04/NOTES.md): plain allocations, so the row is
            # whichever of the two this process was handed
            row["note"] = "plain allocations: one of two placement modes ~6 % apart (profiles/r04/NOTES.md)"
        if leg.name.endswith("_u8"):
            # 2 bytes per input sample: a quarter of the float32 row's input, in the same or less time -- this launch is bound ON the chip (the
            # tile's conversion and its LDS reads), so its fraction of the HBM roofline says little about it
            row["note"] = "u8 input moves a quarter of the float32 row's input bytes: bound on the chip, compare ms with the float32 row of the same ratio"
        if world > 1:
            row["scaling"] = "weak"
            row["workload"] += " -- per rank, %d ranks (weak scaling: the per-GPU launch is the same at every N)" % world
            row["frac_note"] = "per-GPU algorithmic bytes / slowest rank's mean launch time / 8 TB/s"
        other_rows.append(row)

    others = timed_others
    # ---- headline: W untimed + exactly K timed steps
    elapsed, kern_list = time_leg(ctx, head, args.steps, args.warmup)
    ms_per_step = elapsed * 1e3 / args.steps
    job_samples = shard.sum_over_ranks([head.n_gpu], ctx_red)[0]      # all ranks' samples per step
    value = job_samples / (ms_per_step * 1e-3) / 1e6                  # whole-job complex MS/s
    job_bytes = shard.sum_over_ranks([head.bytes_per_launch], ctx_red)[0]
    kern_ms = float(np.mean(kern_list))                               # this rank's launches (rank 0's in the line)
    kern_ms_slowest = shard.max_over_ranks(kern_ms, ctx_red)          # the rank that bounds the job

    # ---- parity (every rank, its own channels) and the cross-rank checksum; the oracle is the checker only
    worst, nwin, W = head.check(full=world > 1)          # N > 1: EVERY local channel, first and last window
    worst_all = shard.max_over_ranks(worst, ctx_red)
    nwin_all = int(shard.sum_over_ranks([nwin], ctx_red)[0])
    csum = shard.sum_over_ranks(checksum(ctx, head), ctx_red)
    is_tx10 = getattr(head, "out_fmt", "f32") == "tx10"
    parity = {"rel_rms_max": worst_all, "windows": nwin_all, "window_len": W, "tol": TOL,
              "ok": bool(worst_all <= (2e-3 if is_tx10 else TOL)), "ranks_checked": world,
              "checked": "last timed step, every rank: " + ("every local channel, first and last window" if world > 1 and head.kind == "fir"
                                                            else "first and last local channel")}
    held = input_held(head)
    if held is not None:
        parity["input_is_the_synthetic_stream"] = bool(held)   # bit for bit, before the timed steps (above) AND after them
        parity["input_checked"] = "before and after the timed steps"
        parity["ok"] = parity["ok"] and bool(held)
    if is_tx10:
        parity["note"] = "10-bit output: figure = fraction of packed bytes differing from the oracle's packing (1-LSB code-boundary flips)"
    for leg, row in zip(others, other_rows):
        w, c, wl_ = leg.check(full=False)
        if world > 1:
            w, c = shard.max_over_ranks(w, ctx_red), int(shard.sum_over_ranks([c], ctx_red)[0])
        row["parity"] = {"rel_rms_max": w, "windows": c, "window_len": wl_, "tol": TOL, "ok": bool(w <= TOL)}
        held = input_held(leg)
        if held is not None:
            row["parity"]["input_is_the_synthetic_stream"] = bool(held)
            row["parity"]["ok"] = row["parity"]["ok"] and bool(held)

    sharded = world > 1 and head.kind == "fir"
    traffic, traffic_stale = pmc_traffic(head.key)
    out = {
        "metric": "complex-float32 MS/s through 256-tap blkconv FIR; % of HBM roofline" if wl == "fir"
        else "complex-float32 input MS/s through libdsp %s; %% of HBM roofline" % wl,
        "value": value, "unit": "MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True,
        # a channel-sharded FIR job is FIXED (64 channels x 2^24 whatever N is): strong scaling;
        # resample/decimate at N > 1 are N replicas of the one-GPU job: weak
        "scaling": "strong" if sharded else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "precondition_s": args.precondition,
        "config": {"workload": head.workload, "channels_per_gpu": head.nch, "samples_per_gpu": head.n_gpu,
                   "sharding": "independent channels per rank, no data-path collective" if world > 1 else "single GPU"},
        # per launch of the dominant kernel.  N > 1: the whole job's algorithmic bytes per step over the
        # slowest rank's mean launch time, against N x 8 TB/s
        "roofline": {"bound": "hbm", "achieved": job_bytes / (kern_ms_slowest * 1e-3) / 1e9, "peak": HBM_PEAK_GBS * world,
                     "unit": "GB/s", "frac": job_bytes / (kern_ms_slowest * 1e-3) / 1e9 / (HBM_PEAK_GBS * world),
                     "traffic": traffic, "kernel": head.kernel, "kernel_ms": kern_ms_slowest,
                     "kernel_ms_min": float(np.min(kern_list)), "kernel_ms_max": float(np.max(kern_list)),
                     "kernel_ms_std": float(np.std(kern_list)), "kernel_ms_median": float(np.median(kern_list)),
                     "algorithmic_bytes_per_launch": head.bytes_per_launch},
        "parity": parity,
        "checksum": {"samples": csum[0], "sum_re": csum[1], "sum_im": csum[2], "sum_abs2": csum[3],
                     "over": "all ranks' last output (float64 sums, all-reduced)"},
    }
    if getattr(head, "variant", None):
        out["roofline"]["variant"] = head.variant     # which data-movement variant this device's measurement picked
    out["config"]["buffers"] = "plain allocations (torch.empty); nothing is selected for speed"
    if getattr(head, "pairs", None):
        out["config"]["buffers"] += "; headline on the median of three plain pairs"
        out["config"]["pairs"] = head.pairs
    if args.telemetry and rank == 0:
        out["roofline"]["telemetry"] = telemetry(ctx, head, kern_ms)
    if traffic_stale:
        out["roofline"]["traffic_stale"] = True      # simplefe_amd/csrc changed since the PMC pass in profiles/
    if world > 1:
        out["roofline"]["note"] = ("achieved = all ranks' algorithmic bytes per step / the slowest rank's mean launch time; "
                                   "peak = %d x 8 TB/s; kernel_ms_min/max/std are rank 0's launches" % world)
    if sharded:
        out["config"]["workload"] = ("%d independent cf32 channels x %s samples, 256-tap FIR (blkconv law), channel-sharded %s per GPU over %d GPUs, "
                                     "device-resident in/out" % (total_channels, _p2(head.n),
                                                                 "%d/%d" % (total_channels, world), world))
        out["config"]["channels_total"] = total_channels
        out["config"]["samples_per_channel"] = head.n
        out["config"]["samples_total"] = int(job_samples)
    if other_rows or other_errors:
        out["other_configs"] = other_rows + other_errors
    if world > 1 and head.kind == "fir" and not np.iscomplexobj(head.taps):
        out["split_stream"] = split_stream_check(ctx, head.taps)
    if sharded and not args.no_others and args.input == "f32" and args.output == "f32" and args.algo == "auto":
        # The same fixed job driven from ONE process through sfe_dsp_fir_group_* (the form a C++ flowgraph
        # uses), outside every timed region: rank 0 drives all N devices while the other ranks wait on the
        # HOST (a gloo group: an RCCL barrier would keep a kernel spinning on the very GPUs being timed).
        host = dist.new_group(backend="gloo")
        torch.cuda.synchronize()
        dist.barrier(group=host)
        if rank == 0:
            try:
                devs = [0] * world if os.environ.get("SFE_BENCH_ONE_DEVICE") else list(range(world))
                g = group_leg(ctx, head.taps, total_channels, head.n, devs, args.other_steps, 3)
                g["workload"] = "the same job, ONE process driving %d device(s) through sfe_dsp_fir_group_* (no torch.distributed)" % world
                g["checksum_equals_ranks"] = bool(all(abs(a - b) <= 1e-9 * max(1.0, abs(b)) for a, b in
                                                      zip([g["checksum"][k] for k in ("samples", "sum_re", "sum_im", "sum_abs2")], csum)))
                out["single_process_group"] = g
            except Exception as e:          # reported, never allowed to take the headline with it
                out["single_process_group"] = {"error": "%s: %s" % (type(e).__name__, e)}
        dist.barrier(group=host)
    if rank == 0 and world == 1 and not args.no_cpu:
        if head.kind == "fir":
            out["cpu_baseline"] = cpu_baseline_fir(np.real(head.taps).astype(np.float32), args.cpu_seconds)
            if np.iscomplexobj(head.taps):
                out["cpu_baseline"]["sample"] += " (real-tap passes; complex taps cost four such passes, not two)"
        else:
            out["cpu_baseline"] = cpu_baseline_rs(wl, head.taps, head.U, head.rate, args.cpu_seconds)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if not parity["ok"]:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
