#!/usr/bin/env python3
"""bench.py -- throughput of the libdsp hot path on MI355X, one JSON line on stdout.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload fir|resample|decimate]

Default line (what the driver records): BASELINE.json configs[1] -- a 256-tap FIR over a
2^28-sample complex-float32 stream, device-resident in and out, one GPU.  A "step" is one pass
of the hot path over that batch: one sfe_dsp_fir_process_stream call (the FFT overlap-save
kernel plus the 2 KiB history carry-over).  For N > 1 (launched by torch.distributed.run, one
rank per GPU) every rank filters its own 8 independent channels of 2^25 samples -- the
channel-sharded configs[4] shape, same per-GPU sample count, no data-path collective; the
only cross-rank traffic is the barrier and the MAX-reduction of the elapsed time (RCCL).

`roofline`     algorithmic bytes of the dominant kernel per launch / its mean duration, timed
               with HIP events on the launch stream inside the timed region; `traffic` is the
               PMC-measured HBM bytes per launch from profiles/ (null if no pass was recorded).
`cpu_baseline` the CPU oracle (oracle/, a port of the reference algorithm; oracle/_ref for
               resample/decimate = the reference's own code) timed on this host on a bounded
               sample of the same workload, rank 0, N = 1 only.  A reported baseline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # ~0.2 s of GPU time; a 20-step run sits on the DVFS transient
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="fir", choices=["fir", "resample", "decimate"])
    ap.add_argument("--log2n", type=int, default=None, help="samples per GPU = 2^log2n (default per workload)")
    ap.add_argument("--algo", default="auto", choices=["auto", "fft", "direct"])
    ap.add_argument("--channels", type=int, default=None,
                    help="channels per GPU (default 1 at N=1, 8 at N>1: the 64-channel config over 8 GPUs)")
    ap.add_argument("--input", default="f32", choices=["f32", "u8"],
                    help="u8: the stream is the device wire format, converted on load (fused RX converter, N2)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def pmc_traffic(workload_key):
    """HBM bytes per launch from the committed PMC summary, if one exists for this workload."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    if not os.path.isdir(pdir):
        return None
    for f in sorted(os.listdir(pdir)):
        if f.startswith("pmc_") and f.endswith(".json"):
            try:
                d = json.load(open(os.path.join(pdir, f)))
            except Exception:
                continue
            if d.get("workload") == workload_key and d.get("hbm_bytes_per_launch"):
                best = d["hbm_bytes_per_launch"]
    return best


def cpu_baseline_fir(taps, n_budget_s):
    """Oracle port of blkconv (fft_len 4096, blk 3841), I and Q as two real passes, 1 thread."""
    from oracle import binding as orc
    from simplefe_amd import synth
    n = 1 << 20
    x = synth.synth_cf32(n)
    xr, xi = np.ascontiguousarray(x[0::2]), np.ascontiguousarray(x[1::2])
    t0 = time.perf_counter()
    orc.Blkconv(taps, 4096).stream(xr)
    orc.Blkconv(taps, 4096).stream(xi)
    dt = time.perf_counter() - t0
    reps = max(1, int(n_budget_s / max(dt, 1e-6)))
    reps = min(reps, 1024)
    cr, ci = orc.Blkconv(taps, 4096), orc.Blkconv(taps, 4096)
    t0 = time.perf_counter()
    for _ in range(reps):
        cr.stream(xr)
        ci.stream(xi)
    dt = time.perf_counter() - t0
    return {"value": reps * n / dt / 1e6, "unit": "MS/s", "cores": 1, "kind": "port",
            "sample": f"{reps} x 2^20 cf32 samples, blkconv port fft_len 4096 (own float32 FFT; "
                      f"FFTW absent), I and Q as two real passes, {dt:.1f} s"}


def cpu_baseline_rs(which, taps, U, rate, n_budget_s):
    """The reference's own resample/decimate class (oracle/_ref) when present, else the port."""
    from oracle import binding as orc
    from simplefe_amd import synth
    use_ref = orc.ref_lib() is not None
    cls = {("resample", True): orc.RefResample, ("resample", False): orc.Resample,
           ("decimate", True): orc.RefDecimate, ("decimate", False): orc.Decimate}[(which, use_ref)]
    n = 1 << 18
    B = 4096
    x = synth.synth_cf32(n)
    xr, xi = np.ascontiguousarray(x[0::2]), np.ascontiguousarray(x[1::2])
    t0 = time.perf_counter()
    cls(taps, U, B).stream(xr, rate)
    cls(taps, U, B).stream(xi, rate)
    dt = time.perf_counter() - t0
    reps = min(max(1, int(n_budget_s / max(dt, 1e-6))), 4096)
    a, b = cls(taps, U, B), cls(taps, U, B)
    t0 = time.perf_counter()
    for _ in range(reps):
        a.stream(xr, rate)
        b.stream(xi, rate)
    dt = time.perf_counter() - t0
    return {"value": reps * n / dt / 1e6, "unit": "MS/s", "cores": 1,
            "kind": "reference" if use_ref else "port",
            "sample": f"{reps} x 2^18 cf32 samples, libdsp {which} class chunked at {B}, "
                      f"I and Q as two real passes, {dt:.1f} s"}


def main():
    args = parse()
    from simplefe_amd import shard
    rank, local_rank, world = shard.env_ranks()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libsfe_dsp has no CPU fallback")
    if os.environ.get("SFE_BENCH_ONE_DEVICE"):      # rehearsal: every rank on cuda:0 (with SFE_DIST_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    shard.init_process_group(dev)            # RCCL when WORLD_SIZE > 1

    from simplefe_amd import api, lib, synth
    L = lib.load()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    wl = args.workload
    if wl == "fir":
        log2n = args.log2n or 28
        nch = args.channels or (1 if world == 1 else 8)   # N > 1: 8*N channels, block-partitioned over ranks
        taps = synth.taps_cfg2()
        workload = ("256-tap FIR (blkconv law), 2^%d cf32 samples per GPU, %d channel(s) x 2^%d, "
                    "device-resident in/out" % (log2n, nch, log2n - (nch.bit_length() - 1)))
        key = "fir256_cf32_2p%d%s" % (log2n, "_direct" if args.algo == "direct" else "")
    elif wl == "resample":
        log2n = args.log2n or 28
        nch = 1
        taps = synth.taps_cfg3()
        U, rate = 3, 5.0 / 3.0
        workload = "rational resample 5/3, 381-tap prototype (127 per arm), 2^%d cf32 in" % log2n
        key = "resample5o3_cf32_2p%d" % log2n
    else:
        log2n = args.log2n or 30
        nch = 1
        taps = synth.taps_cfg4()
        U, rate = 1, 8.0
        workload = "decimate by 8, 64-tap anti-alias FIR, 2^%d cf32 in" % log2n
        key = "decimate8_cf32_2p%d" % log2n
    n_gpu = 1 << log2n
    n = n_gpu // nch                      # samples per channel
    stream = torch.cuda.current_stream().cuda_stream

    # ---- device-resident synthetic input (generated on the GPU; host twin: synth.py)
    x = torch.empty(nch * n * 2, dtype=torch.float32, device=dev)
    for c in range(nch):
        gch = shard.channel_block(nch * world, world, rank)[0] + c      # global channel id = its seed
        api.check(L.sfe_dsp_synth_fill(x.data_ptr() + c * n * 8, 2 * n, synth.SEED, gch, 0, stream))
    in_bytes = 8.0
    if args.input == "u8":
        # wire format: (I,Q) byte pairs.  Derived from the float stream so the parity legs below
        # still have a float twin: b = round(127 x) + 128, i.e. x_u8 = (b - 128)/127.
        xb = (torch.round(x * 127.0) + 128.0).clamp_(0, 255).to(torch.uint8)
        x = ((xb.to(torch.float32) - 128.0) * (1.0 / 127.0))          # the float twin (exact)
        in_bytes = 2.0
        key += "_u8"
        workload += ", u8 (I,Q) wire-format input converted on load"
    if wl == "fir":
        n_out = n
        y = torch.empty(nch * n * 2, dtype=torch.float32, device=dev)
        obj = api.Fir(taps, data_complex=True, n_channels=nch, device=local_rank,
                      algo={"auto": lib.FIR_ALGO_AUTO, "fft": lib.FIR_ALGO_FFT, "direct": lib.FIR_ALGO_DIRECT}[args.algo])
        bytes_per_launch = (in_bytes + 8.0) * n_gpu   # 8 B (2 B for u8) read + 8 B written per sample
        kernel = "fir_fft4096_kernel" if args.algo != "direct" else "poly_tiled_kernel"
        src = x
        if args.input == "u8":
            obj.set_input_format(lib.FMT_U8)
            src = xb

        def step():
            obj.process_stream(src.data_ptr(), y.data_ptr(), n, stream=stream)
    else:
        out_cap = int(n / rate) + 8
        y = torch.empty(nch * out_cap * 2, dtype=torch.float32, device=dev)
        obj = api.Rs(taps, U, 4096, mode=lib.RS_RESAMPLE if wl == "resample" else lib.RS_DECIMATE,
                     data_complex=True, n_channels=nch, device=local_rank)
        n_out_box = [0]
        # the library picks the transform-domain kernel for long filters on cf32 streams (api.hip: get_fft_plan)
        kernel = "poly_fft256_kernel" if wl == "resample" and os.environ.get("SFE_RS_FFT", "") != "0" else "poly_tiled_kernel"
        src = x
        if args.input == "u8":
            obj.set_input_format(lib.FMT_U8)
            src = xb

        def step():
            n_out_box[0] = obj.process_stream(src.data_ptr(), n, y.data_ptr(), out_cap, rate, stream=stream)
        step()
        n_out = n_out_box[0]
        bytes_per_launch = in_bytes * n_gpu + 8.0 * n_out * nch
        rs_parity = [float("nan")]
        if rank == 0:
            # parity of the first (fresh-state) pass on windows: output k sits at upsampled
            # position k*S; an oracle object started at input index a0 (a0*U a multiple of S)
            # reproduces outputs k >= a0*U/S + ceil(plen*U/S) exactly-in-law (finite memory).
            from oracle import binding as orc
            S = int(round(rate * U))
            g = int(np.gcd(S, U))
            per = S // g                       # input samples per phase period
            plen = (len(taps) + U - 1) // U + 1
            W = 1 << 13
            worst = 0.0
            for k0 in (0, n_out // 2, n_out - W - 8):
                nin0 = (k0 * S) // U           # input index of output k0
                a0 = max(0, ((nin0 - plen - per) // per) * per)
                j0 = k0 - a0 * U // S          # index of output k0 in the oracle's own stream
                n_span = ((k0 + W) * S) // U + 2 - a0
                n_span = min(n_span, n - a0)
                seg = x[2 * a0: 2 * (a0 + n_span)].cpu().numpy()
                got = y[2 * k0: 2 * (k0 + W)].cpu().numpy()
                for part in (0, 1):
                    ref, _ = orc.Decimate(taps, U, 4096).stream(np.ascontiguousarray(seg[part::2]), rate)
                    m = min(W, len(ref) - j0)
                    worst = max(worst, synth.rel_rms(got[part::2][:m], ref[j0:j0 + m]))
            rs_parity[0] = worst

    for _ in range(args.warmup):
        step()
    timers = [api.Timer() for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        timers[k].start(stream)
        step()
        timers[k].stop(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = shard.max_over_ranks(elapsed, dev)
    ms_per_step = elapsed * 1e3 / args.steps
    kern_ms = float(np.mean([t.elapsed_ms() for t in timers]))
    value = world * n_gpu / (ms_per_step * 1e-3) / 1e6      # whole-job complex MS/s

    # ---- parity on windows of the LAST step's output (oracle is the checker only)
    parity = None
    if rank == 0:
        from oracle import binding as orc
        W = 1 << 13
        worst = 0.0
        if wl == "fir":
            hlen = len(taps) - 1
            starts = [0, 3840 - 100, n // 2 - 77, n - W]
            for s0 in starts:
                s0 = max(0, min(s0, n - W))
                lo = s0 - hlen
                if lo >= 0:
                    seg = x[2 * lo: 2 * (s0 + W)].cpu().numpy()
                else:   # history = tail of the same buffer fed in the previous step
                    seg = np.concatenate([x[2 * (n + lo): 2 * n].cpu().numpy(), x[: 2 * (s0 + W)].cpu().numpy()])
                got = y[2 * s0: 2 * (s0 + W)].cpu().numpy()
                for part in (0, 1):
                    ref = orc.Blkconv(taps, 4096).stream(np.ascontiguousarray(seg[part::2]))[hlen:]
                    worst = max(worst, synth.rel_rms(got[part::2], ref))
        else:
            worst = rs_parity[0]
        parity = {"rel_rms_max": worst, "windows": 4 if wl == "fir" else 3, "window_len": W if wl == "fir" else 8192,
                  "tol": 1e-5, "ok": bool(worst <= 1e-5)}

    out = {
        "metric": "complex-float32 MS/s through 256-tap blkconv FIR; % of HBM roofline" if wl == "fir"
        else "complex-float32 input MS/s through libdsp %s; %% of HBM roofline" % wl,
        "value": value, "unit": "MS/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "channels_per_gpu": nch, "samples_per_gpu": n_gpu,
                   "sharding": "independent channels per rank, no data-path collective" if world > 1 else "single GPU"},
        "roofline": {"bound": "hbm", "achieved": bytes_per_launch / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": bytes_per_launch / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic": pmc_traffic(key), "kernel": kernel, "kernel_ms": kern_ms,
                     "algorithmic_bytes_per_launch": bytes_per_launch},
    }
    if parity is not None:
        out["parity"] = parity
    if rank == 0 and world == 1 and not args.no_cpu:
        if wl == "fir":
            out["cpu_baseline"] = cpu_baseline_fir(taps, args.cpu_seconds)
        else:
            out["cpu_baseline"] = cpu_baseline_rs(wl, taps, U, rate, args.cpu_seconds)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
