import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from simplefe_amd import build, lib
lib.LIB_PATH = build.build_lib(diag=True)
from simplefe_amd import api, synth
n = (1 << 24) + 12345
x = api.DeviceArray(2 * n); x.fill_synth(synth.SEED)
f = api.Fir(synth.taps_cfg2(), data_complex=True, algo=lib.FIR_ALGO_FFT)
outs = {}
for q in ("0", "1", "2", "3", "5"):
    os.environ["SFE_FIR_VARIANT"] = "X"; os.environ["SFE_FIR_TQS"] = q
    y = api.DeviceArray(2 * n); y.zero()
    f.reset()
    for rep in range(3):          # counters must come back to zero between launches
        f.reset(); f.process_stream(x, y, n)
    outs[q] = y.to_numpy()
    print(q, "equal to tqs=0:", np.array_equal(outs[q], outs["0"]))
