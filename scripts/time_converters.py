import sys
sys.path.insert(0, "/root/repo")
from simplefe_amd import api, lib, synth
L = lib.load()
n = 1 << 30     # floats
x = api.DeviceArray(n); x.fill_synth(synth.SEED)
o = api.DeviceArray(n * 5 // 16 + 8)
u = api.DeviceArray(n // 4)
for name, fn, bytes_ in (("tx_f32_to_10bit", lambda: L.sfe_dsp_tx_f32_to_10bit(x.ptr, o.ptr, n, None), n * 4 + n * 5 // 4),
                         ("rx_u8_to_f32", lambda: L.sfe_dsp_rx_u8_to_f32(u.ptr, x.ptr, n, None), n + n * 4)):
    ts = []
    for it in range(20):
        t = api.Timer(); t.start(); lib.check(fn()); t.stop(); ts.append(t.elapsed_ms())
    m = sum(ts[5:]) / 15
    print(f"{name}: {m:.3f} ms  {bytes_ / m / 1e9:.2f} TB/s")
