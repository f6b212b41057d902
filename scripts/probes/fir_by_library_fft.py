"""Context number: the same overlap-save FIR composed from library pieces (torch.fft on rocFFT:
strided segment view -> batched 4096-point FFT -> spectrum multiply -> inverse FFT -> slice), on the
2^28-sample cf32 stream of the bench.  Not part of the product or of any test."""
import torch

dev = torch.device("cuda")
n, N, taps = 1 << 28, 4096, 256
adv = N - taps
x = torch.randn(n + N, dtype=torch.complex64, device=dev)
h = torch.randn(taps, dtype=torch.float32, device=dev)
H = torch.fft.fft(torch.nn.functional.pad(h, (0, N - taps)).to(torch.complex64))
nseg = n // adv
y = torch.empty(nseg * adv, dtype=torch.complex64, device=dev)


def run():
    seg = x.as_strided((nseg, N), (adv, 1))
    Y = torch.fft.ifft(torch.fft.fft(seg, dim=1) * H, dim=1)
    y.view(nseg, adv).copy_(Y[:, taps:])


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(10):
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
print(f"library-FFT overlap-save FIR, 2^28 cf32, 256 taps: median {ts[5]:.2f} ms (min {ts[0]:.2f})")
