"""RCCL itself on the GPU box (`-m gpu`): a world-size-1 "nccl" process group on cuda:0 running
the control-plane collectives bench.py uses at N > 1 (shard.max_over_ranks / sum_over_ranks /
barrier, all_gather_into_tensor) on DEVICE tensors, and the point-to-point span-tail exchange
(shard.halo_exchange: batched RCCL send/recv, here looped back to the same rank) feeding
sfe_dsp_fir_load_history.  One GPU cannot host two RCCL ranks, so this is what can run here; the
two-rank logic runs over gloo (tests/test_shard_gloo.py, tests/test_gpu_bench.py).  The first
8-GPU run is then not the first time librccl is loaded and its kernels launched next to ours."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch
import torch.distributed as dist
from simplefe_amd import api, lib, shard, synth

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
rank, local_rank, world = shard.init_process_group(dev, force=True)
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1

# the control-plane collectives of bench.py, on device tensors
assert shard.max_over_ranks(3.5, dev) == 3.5
assert shard.sum_over_ranks([1.0, -2.0, 2.0 ** 40], dev) == [1.0, -2.0, 2.0 ** 40]
shard.barrier()
t = torch.arange(512, dtype=torch.float32, device=dev)
g = torch.zeros(512, dtype=torch.float32, device=dev)
dist.all_gather_into_tensor(g, t)
torch.cuda.synchronize()
assert torch.equal(g, t)

# the data-path exchange of a cut stream: the span tail by RCCL send/recv (looped back), then into
# the library as carried history -- the second span must equal the one-handle stream bit for bit
L = lib.load()
stream = torch.cuda.current_stream().cuda_stream
taps = synth.taps_cfg2()
HL, cut, n = 256, 3840 * 40, 3840 * 40 + 100000
x = torch.empty(2 * n, dtype=torch.float32, device=dev)
api.check(L.sfe_dsp_synth_fill(x.data_ptr(), 2 * n, synth.SEED, 5, 0, stream))
torch.cuda.synchronize()
one = api.Fir(taps, data_complex=True)
y_one = torch.empty_like(x)
one.process_stream(x.data_ptr(), y_one.data_ptr(), n, stream=stream)
halo = shard.halo_exchange(x[: 2 * cut], 2 * HL, left=0, right=0)        # RCCL send + recv kernels
assert halo.is_cuda and torch.equal(halo, x[2 * (cut - HL): 2 * cut])
span = api.Fir(taps, data_complex=True)
span.load_history(halo.data_ptr(), HL, stream=stream)
y_span = torch.empty(2 * (n - cut), dtype=torch.float32, device=dev)
span.process_stream(x.data_ptr() + 8 * cut, y_span.data_ptr(), n - cut, stream=stream)
torch.cuda.synchronize()
assert torch.equal(y_span, y_one[2 * cut:]), "span after an RCCL-carried halo differs from the uncut stream"
assert shard.halo_from_left(x, 2 * HL).abs().max().item() == 0.0         # a world of one has no left neighbour
dist.barrier()
# "nccl" on ROCm IS RCCL: the worker has librccl mapped next to libsfe_dsp
maps = open("/proc/self/maps").read()
assert "librccl" in maps and "libsfe_dsp.so" in maps
dist.destroy_process_group()
print("RCCL_OK", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "", flush=True)
"""


def test_world_size_1_rccl_group_runs_the_collectives_and_the_p2p_halo(tmp_path):
    script = tmp_path / "rccl_worker.py"
    script.write_text(WORKER % {"root": ROOT})
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("SFE_DIST_BACKEND", None)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout + r.stderr
