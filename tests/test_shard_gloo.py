"""The N > 1 path on CPU: two and EIGHT gloo ranks (the driver's N = 8 shape; the GPU box allows at most
six processes on its card, so eight ranks are rehearsed here), channel-block sharding, MAX/SUM control
collectives, rank 0's choice handed to every rank, the host-side (gloo) barrier group.
The per-channel "filter" here is the oracle (this is a test); on GPUs bench.py runs the same
sharding logic with libsfe_dsp kernels and RCCL."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
import pytest
import torch
from simplefe_amd import shard, synth
from oracle import binding as orc
rank, local_rank, world = shard.init_process_group(torch.device("cpu"))
NCH, N = int(os.environ.get("SFE_TEST_NCH", "6")), 4000
first, count = shard.channel_block(NCH, world, rank)
# bench.py at N > 1: rank 0 measures the kernel variant, every rank runs what rank 0 chose
chosen = 2 if rank == 0 else 0
chosen = int(round(shard.sum_over_ranks([float(chosen) if rank == 0 else 0.0])[0]))
assert chosen == 2
# ... and the host-side barrier group the single-process leg waits on
import torch.distributed as dist
host = dist.new_group(backend="gloo")
dist.barrier(group=host)
taps = synth.taps_cfg2()
t0 = time.perf_counter()
acc = [0.0, 0.0, 0.0]
for c in range(first, first + count):
    x = synth.synth_cf32(N, ch=c)
    for part in (0, 1):
        y = orc.Blkconv(taps, 1024).stream(np.ascontiguousarray(x[part::2])).astype(np.float64)
        acc[0] += float(y.sum()); acc[1] += float((y * y).sum()); acc[2] += len(y)
el = time.perf_counter() - t0 + 0.01 * rank
shard.barrier()
tot = shard.sum_over_ranks(acc)
mx = shard.max_over_ranks(el)
assert mx >= el
if rank == 0:
    print("RESULT", world, tot[0], tot[1], int(tot[2]), flush=True)
import torch.distributed as dist
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_channel_block_partition():
    from simplefe_amd import shard
    for nch in (1, 6, 8, 64, 65):
        for world in (1, 2, 3, 4, 8):
            blocks = [shard.channel_block(nch, world, r) for r in range(world)]
            flat = [c for f, k in blocks for c in range(f, f + k)]
            assert flat == list(range(nch))
            assert max(k for _, k in blocks) - min(k for _, k in blocks) <= 1



@pytest.mark.parametrize("ranks,nch", [(2, 6), (8, 19)])
def test_gloo_channel_sharding_matches_single_process(tmp_path, ranks, nch):
    from oracle import binding as orc
    from simplefe_amd import synth
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", SFE_TEST_NCH=str(nch))
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
        capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0].split()
    world, s1, s2, cnt = int(line[1]), float(line[2]), float(line[3]), int(line[4])
    assert world == ranks and cnt == nch * 2 * 4000
    taps = synth.taps_cfg2()
    e1 = e2 = 0.0
    for c in range(nch):
        x = synth.synth_cf32(4000, ch=c)
        for part in (0, 1):
            y = orc.Blkconv(taps, 1024).stream(np.ascontiguousarray(x[part::2])).astype(np.float64)
            e1 += float(y.sum())
            e2 += float((y * y).sum())
    assert abs(s1 - e1) <= 1e-9 * max(1.0, abs(e1)) and abs(s2 - e2) <= 1e-9 * e2


# ------------------------------------------------------------- one stream cut across ranks
SPLIT_WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import pytest
import torch
import torch.distributed as dist
from simplefe_amd import shard, synth
from oracle import binding as orc
rank, local_rank, world = shard.init_process_group(torch.device("cpu"))
N, HL = 50000, 255
taps = synth.taps_cfg2()
first, count = shard.span_block(N, world, rank, quantum=3840)
x = synth.synth_f32(count, first=first)                      # this rank's span of the one real stream
halo = shard.halo_from_left(torch.from_numpy(x), HL).numpy()  # the n_taps-1 samples before the span (zeros on rank 0)
y = orc.Blkconv(taps, 4096).stream(np.concatenate([halo, x]))[HL:]   # the rank's filter, state from the halo
# gather the spans on rank 0 for the check (test only; the product keeps outputs on the owning GPU)
parts = [None] * world
dist.all_gather_object(parts, (first, y))
if rank == 0:
    full = np.concatenate([p[1] for p in sorted(parts, key=lambda p: p[0])])
    ref = orc.Blkconv(taps, 4096).stream(synth.synth_f32(N))
    print("RESULT", world, len(full), synth.rel_rms(full, ref), flush=True)
dist.destroy_process_group()
"""


def test_span_block_partition():
    from simplefe_amd import shard
    for n in (1, 3839, 3840, 50000, 2 ** 20 + 17):
        for world in (1, 2, 3, 8):
            for q in (1, 5, 3840):
                spans = [shard.span_block(n, world, r, q) for r in range(world)]
                assert sum(c for _, c in spans) == n
                pos = 0
                for f, c in spans:
                    assert f == pos or c == 0
                    assert c == 0 or f % q == 0
                    pos += c


@pytest.mark.parametrize("ranks", [2, 8])
def test_gloo_single_stream_split_with_halo_exchange(tmp_path, ranks):
    """SURVEY 8(e) row 3 rehearsed on CPU: one stream, two / eight ranks, cut on transform boundaries, the
    255-sample halo sent point-to-point to the right neighbour; the stitched result equals the
    uncut stream's (the oracle stands in for the kernel here -- on GPUs the same shard.* calls
    feed sfe_dsp_fir_load_history, tests/test_gpu_split.py)."""
    script = tmp_path / "split_worker.py"
    script.write_text(SPLIT_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
         "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
        capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert int(line[1]) == ranks and int(line[2]) == 50000 and float(line[3]) <= 1e-6
