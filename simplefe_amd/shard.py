"""Channel sharding across the GPUs of one node.

Channels are independent streams (one reference object each), so the path shards with no
data-path exchange: rank r owns a contiguous block of channels and keeps its outputs resident.
The only collectives are the control-plane ones below -- a MAX over ranks of the elapsed time
and a SUM of per-rank checksums -- over RCCL ("nccl") on GPUs or gloo on CPU-only hosts.
"""
import os


def channel_block(n_channels, world, rank):
    """(first, count) of the contiguous channel block rank `rank` owns."""
    base, extra = divmod(int(n_channels), int(world))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def span_block(n_samples, world, rank, quantum=1):
    """(first, count) of the contiguous span of ONE stream that rank `rank` filters when the stream is
    cut across `world` GPUs.  Cuts fall on multiples of `quantum` samples: with
    quantum = the FIR transform advance (3840 for <= 257 taps) the spans' outputs are bit-identical
    to the uncut stream's; any quantum is correct to rounding.  The last rank takes the ragged end."""
    n, world, q = int(n_samples), int(world), max(1, int(quantum))
    per = ((n + world - 1) // world + q - 1) // q * q
    first = min(rank * per, n)
    return first, max(0, min(per, n - first))


def halo_exchange(tail, halo_len, left, right):
    """Send the last `halo_len` elements of `tail` to rank `right` and receive as many from rank
    `left` (either may be None: nothing sent / zeros returned).  One batched RCCL send/recv pair --
    point to point over one xGMI link, not a collective; under gloo the same calls on host tensors.
    A rank may name itself on both sides (a world of one: the loop-back the one-GPU box can run)."""
    import torch
    import torch.distributed as dist
    halo = torch.zeros(halo_len, dtype=tail.dtype, device=tail.device)
    send = tail[-halo_len:].contiguous() if tail.numel() >= halo_len else torch.cat(
        [torch.zeros(halo_len - tail.numel(), dtype=tail.dtype, device=tail.device), tail])
    ops = []
    if right is not None:
        ops.append(dist.P2POp(dist.isend, send, right))
    if left is not None:
        ops.append(dist.P2POp(dist.irecv, halo, left))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        # under RCCL wait() only orders torch's current stream behind the transfer; the library's
        # handles run on streams of their own, so make the halo visible to every stream
        if halo.is_cuda:
            torch.cuda.current_stream(halo.device).synchronize()
    return halo


def halo_from_left(tail, halo_len, device=None):
    """The one data exchange a cut stream needs: every rank sends the last `halo_len` samples of its
    span to its right neighbour (n_taps-1 samples, 2 KiB for 256 taps cf32) and receives its own
    halo from the left; rank 0 gets zeros (stream start).  `tail`: 1-D float32 tensor holding AT
    LEAST the span's last halo_len elements (on the GPU under RCCL, on the host under gloo)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return torch.zeros(halo_len, dtype=tail.dtype, device=tail.device)
    rank, world = dist.get_rank(), dist.get_world_size()
    return halo_exchange(tail, halo_len, rank - 1 if rank > 0 else None, rank + 1 if rank + 1 < world else None)


def env_ranks():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(device=None, force=False):
    """One process per GPU, rendezvous on 127.0.0.1; returns (rank, local_rank, world).  A single
    rank makes no group unless `force` (the world-size-1 RCCL group of tests/test_gpu_rccl.py)."""
    import torch.distributed as dist
    rank, local_rank, world = env_ranks()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # RCCL ("nccl") on GPUs, gloo on CPU-only hosts; SFE_DIST_BACKEND=gloo lets the N > 1 code
        # path be rehearsed with several ranks on ONE GPU (RCCL refuses two ranks on a device)
        backend = os.environ.get("SFE_DIST_BACKEND") or ("nccl" if device is not None and device.type == "cuda" else "gloo")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def max_over_ranks(value, device=None):
    """MAX of one float over the ranks.  Runs the collective whenever a process group exists (a
    world-size-1 RCCL group included: tests/test_gpu_rccl.py drives exactly this code on the one-GPU
    box); without a group it is the identity."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values, device=None):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
