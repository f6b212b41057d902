"""bench.py as the driver runs it, on the GPU box (`-m gpu`): the N > 1 launch path from a bare
shell, per-rank parity, the cross-rank checksum, and the shape of the one JSON line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                       env=env, timeout=timeout, cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus2_from_a_bare_shell_starts_two_ranks_and_checks_both():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must start 2 ranks itself (here both on the
    one device, over gloo: SFE_BENCH_ONE_DEVICE=1) and run BASELINE configs[4] as written: the
    FIXED 64-channel job block-partitioned 32 + 32 (strong scaling), every local channel of every
    rank checked; the all-reduced checksum equals the one-process run over the same 64 channels
    (seeds 0..63) -- the channel partition changes nothing."""
    r2, j2 = _bench("--gpus", "2", "--log2n", "24", "--steps", "3", "--warmup", "1", "--no-cpu",
                    env_extra={"SFE_BENCH_ONE_DEVICE": "1"})
    assert r2.returncode == 0, r2.stdout + r2.stderr
    assert j2["n_gpus"] == 2 and j2["scaling"] == "strong"
    cfg = j2["config"]
    assert cfg["channels_per_gpu"] == 32 and cfg["channels_total"] == 64 and cfg["samples_per_channel"] == 1 << 18
    assert "64 independent cf32 channels" in cfg["workload"] and "64/2 per GPU" in cfg["workload"]
    assert j2["parity"]["ok"] and j2["parity"]["ranks_checked"] == 2
    assert j2["parity"]["windows"] == 64 * 2                   # every channel: first and last window
    assert j2["checksum"]["samples"] == 1 << 24
    assert j2["roofline"]["peak"] == 16000.0 and 0 < j2["roofline"]["frac"] < 1
    for k in ("kernel_ms_min", "kernel_ms_max", "kernel_ms_std"):
        assert k in j2["roofline"]
    # one stream cut across the two ranks, halo exchanged point to point: the seam matches the uncut stream
    ss = j2["split_stream"]
    assert ss["ok"] and ss["spans"] == 2 and ss["exchange"].startswith("point-to-point"), ss
    r1, j1 = _bench("--gpus", "1", "--log2n", "24", "--channels", "64", "--steps", "3", "--warmup", "1", "--no-cpu")
    assert r1.returncode == 0, r1.stdout + r1.stderr
    assert j1["n_gpus"] == 1 and j1["checksum"]["samples"] == j2["checksum"]["samples"]
    assert j1["config"]["channels_per_gpu"] == 64
    for k in ("sum_re", "sum_im", "sum_abs2"):
        a, b = j1["checksum"][k], j2["checksum"][k]
        assert abs(a - b) <= 1e-9 * max(1.0, abs(a)), (k, a, b)


def test_bench_gpus4_rehearsal_and_the_single_process_group_form():
    """Four ranks on the one device (the card allows six processes; eight ranks are rehearsed over gloo on
    the CPU, tests/test_shard_gloo.py): 64 channels as 16 + 16 + 16 + 16, one stream cut into four spans,
    rank 0's calibrated variant run by every rank, and -- outside the timed region -- the SAME job driven by
    rank 0 alone through sfe_dsp_fir_group_* over four blocks: same checksum as the four ranks' all-reduce."""
    r4, j4 = _bench("--gpus", "4", "--log2n", "24", "--steps", "3", "--warmup", "1", "--other-steps", "3", "--no-cpu",
                    env_extra={"SFE_BENCH_ONE_DEVICE": "1"}, timeout=900)
    assert r4.returncode == 0, r4.stdout + r4.stderr
    assert j4["n_gpus"] == 4 and j4["scaling"] == "strong" and j4["config"]["channels_per_gpu"] == 16
    assert j4["parity"]["ok"] and j4["parity"]["ranks_checked"] == 4 and j4["parity"]["windows"] == 128
    assert j4["roofline"]["variant"]["chosen_by"].endswith("broadcast to all ranks")
    ss = j4["split_stream"]
    assert ss["ok"] and ss["spans"] == 4 and ss["exchange"].startswith("point-to-point"), ss
    g = j4["single_process_group"]
    assert "error" not in g, g
    assert g["parity"]["ok"] and g["checksum_equals_ranks"] and g["devices"] == [0, 0, 0, 0]
    assert [c for _, _, c in g["shards"]] == [16, 16, 16, 16]
    # and the bare single-process form of the line
    r1, j1 = _bench("--gpus", "8", "--single-process", "--log2n", "24", "--steps", "3", "--warmup", "1",
                    env_extra={"SFE_BENCH_ONE_DEVICE": "1"})
    assert r1.returncode == 0, r1.stdout + r1.stderr
    assert j1["n_gpus"] == 8 and j1["parity"]["ok"] and len(j1["config"]["shards"]) == 8
    # ... and ONE handle over all 64 channels on the one device: the partition into 4 ranks or 8 blocks changes nothing
    r0, j0 = _bench("--gpus", "1", "--log2n", "24", "--channels", "64", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-others")
    assert r0.returncode == 0, r0.stdout + r0.stderr
    for k in ("sum_re", "sum_im", "sum_abs2"):
        a, b, c = j1["checksum"][k], j4["checksum"][k], j0["checksum"][k]
        assert abs(a - b) <= 1e-9 * max(1.0, abs(a)), (k, a, b)
        assert abs(a - c) <= 1e-9 * max(1.0, abs(a)), (k, a, c)


def test_bench_gpus2_default_shape_carries_the_weak_row():
    """No --log2n/--channels at N > 1: the 64 x 2^24 job as the headline (32 channels per rank here)
    and the round-2 weak shape (8 channels x 2^25 per rank) as an other_configs row, both checked."""
    r, j = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--other-steps", "3", "--no-cpu",
                  env_extra={"SFE_BENCH_ONE_DEVICE": "1"}, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert j["config"]["samples_per_channel"] == 1 << 24 and j["config"]["channels_per_gpu"] == 32
    assert j["config"]["samples_total"] == 1 << 30 and j["parity"]["ok"]
    rows = j["other_configs"]
    assert len(rows) == 1 and rows[0]["scaling"] == "weak" and rows[0]["parity"]["ok"], rows
    assert "8 channel(s) x 2^25" in rows[0]["workload"]


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """Never an n_gpus=1 line for --gpus N: a mismatching WORLD_SIZE is an error, not a fallback."""
    r, j = _bench("--gpus", "4", "--no-cpu", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and j is None


def test_bench_default_line_small():
    """The default workload at a reduced size: headline keys, roofline, parity, checksum present."""
    r, j = _bench("--log2n", "24", "--steps", "5", "--warmup", "2", "--cpu-seconds", "2")
    assert r.returncode == 0, r.stdout + r.stderr
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "checksum"):
        assert k in j, k
    assert j["parity"]["ok"] and j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] < 1
    rf = j["roofline"]
    assert rf["kernel_ms_min"] <= rf["kernel_ms_median"] <= rf["kernel_ms_max"] and rf["kernel_ms_std"] >= 0
    assert rf["traffic"] is None or not rf.get("traffic_stale")      # never stale bytes
    # round 5: plain allocations, nothing selected on the measured quantity; the input verified before and after the timed steps
    assert j["config"]["buffers"].startswith("plain allocations")
    assert j["parity"]["input_is_the_synthetic_stream"] is True and j["parity"]["input_checked"] == "before and after the timed steps"
    for row in j.get("other_configs", []):
        assert "error" in row or "buffers" not in row
    cb = j["cpu_baseline"]
    assert cb["cores"] == 1 and cb["all_cores"]["cores"] == cb["host_cores"] >= 1


def test_bench_driver_shape_line_carries_every_row_checked():
    """The driver's own command shape at full size (fewer steps): the headline and EVERY other_configs row -- the BASELINE configs, the general
    rate, and round 5's rows for the kernels of poly_rt_dma.hip: a complex decimation outside the compiled tables, a REAL float32 stream
    (libdsp's native type) through the register-window kernel, the u8 receive wire format through the LDS-DMA fetch -- parity-checked against
    the oracle on the box that timed it; no row missing, none in error."""
    r, j = _bench("--gpus", "1", "--steps", "5", "--warmup", "2", "--no-cpu", timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert j["parity"]["ok"]
    rows = j["other_configs"]
    assert all("error" not in row for row in rows), rows
    assert len(rows) == 11 and all(row["parity"]["ok"] for row in rows), [row.get("workload") for row in rows]
    kernels = {row["kernel"] for row in rows}
    assert {"poly_fft256_kernel", "poly_tiled_kernel", "poly_gen4096_kernel", "poly_rt1_kernel", "poly_rt_dma_kernel", "poly_int4_dma_kernel",
            "fir_fft4096_kernel"} <= kernels, kernels
    assert any("REAL float32 stream" in row["workload"] for row in rows) and any("u8 wire-format input" in row["workload"] for row in rows)
