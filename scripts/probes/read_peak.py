"""Read-side HBM ceiling on this box: torch reductions over 8 GiB (what a decimator mostly does)."""
import torch
x = torch.empty(1 << 31, dtype=torch.float32, device="cuda").normal_()
for name, fn in (("sum f32", lambda: x.sum()), ("max f32", lambda: x.max()), ("view f64 sum", lambda: x.view(torch.float64).sum())):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(20):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    print(f"{name}: median {ts[10]:.3f} ms -> {x.numel() * 4 / ts[10] / 1e9:.2f} TB/s (min {ts[0]:.3f})", flush=True)
