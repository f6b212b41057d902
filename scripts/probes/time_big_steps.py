import os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from simplefe_amd import api, lib, synth
n = 1 << 28
x = api.DeviceArray(2 * n); x.fill_synth(synth.SEED)
t = api.Timer()
print("shape U step ms frac")
for name, U, step, lp in [("/64",1,64,32),("/100",1,100,32),("/128",1,128,32),("/250",1,250,32),("/100 128taps",1,100,128),("10/9",9,10,32),("16/15",15,16,32),("25/24",24,25,32),("9/10",10,9,32),("x16",16,1,32),("x32",32,1,32),("x32 127 taps",32,1,127),("33/32",32,33,32),("147/160",160,147,32),("160/147",147,160,32),("/7 (ref)",1,7,32)]:
    rate = float(np.float32(step)/np.float32(U))
    n = (1 << 28) >> (4 if U >= 8 * step else 0)      # strong interpolation: a sixteenth of the input
    taps = synth.lowpass_taps(lp*U, 0.9*min(1.0/U,1.0/step), gain=float(U))
    cap = n*U//step + 64
    y = api.DeviceArray(2*cap)
    r = api.Rs(taps, U, 4096, mode=lib.RS_RESAMPLE, data_complex=True)
    for _ in range(20): k = r.process_stream(x, n, y, cap, rate)
    v=[]
    for _ in range(5):
        t.start()
        for _ in range(3): r.process_stream(x, n, y, cap, rate)
        t.stop(); v.append(t.elapsed_ms()/3)
    ms=float(np.median(v)); gb=8.0*(n+k)/1e9
    print(f"{name:14s} {U:2d} {step:4d} {ms:9.4f} {gb/ms/8.0:6.3f}" + ("   (2^24 in)" if n < (1 << 28) else ""), flush=True)
    r.close(); y.free()
