#!/usr/bin/env python3
"""Probe: plain device-to-device copy rate of 2 GiB on this box (torch), the ceiling any
16-B/sample kernel is bounded by."""
import torch, time
x = torch.empty(1 << 29, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
for _ in range(5): y.copy_(x)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for _ in range(20): y.copy_(x)
ev1.record(); torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1) / 20
print(f"copy 2 GiB: {ms:.4f} ms -> {2 * x.numel() * 4 / ms / 1e6:.0f} GB/s (read+write)")
