"""One handle per host thread, each on a stream of its own, all running at once -- how a GNU Radio flowgraph drives
the blocks (one scheduler thread per block, SURVEY.md 8(b) "Threading": objects are not re-entrant, callers serialise
PER OBJECT).  What is shared between handles is process-wide: the FIR variant calibrations (a mutex-guarded cache: two
threads with the same shape race for the same entry), the per-device CU count, the thread-local error string.  Every
thread's stream must equal what the same handle produces alone, bit for bit."""
import threading

import numpy as np
import pytest

from simplefe_amd import synth

pytestmark = pytest.mark.gpu


def _jobs(api, L):
    n_fir = 1 << 25                        # 8739 transforms: a first call of this size measures the variants
    n_rs = 1 << 24
    return [
        ("fir a", lambda: api.Fir(synth.taps_cfg2(), data_complex=True), n_fir, 1.0, 3),
        ("fir b", lambda: api.Fir(synth.taps_cfg2(), data_complex=True), n_fir, 1.0, 4),        # same calibration key as "fir a"
        ("decimate", lambda: api.Rs(synth.taps_cfg4(), 1, 4096, mode=L.RS_DECIMATE, data_complex=True), n_rs, 8.0, 5),
        ("resample", lambda: api.Rs(synth.taps_cfg3(), 3, 4096, mode=L.RS_RESAMPLE, data_complex=True), n_rs, 5.0 / 3.0, 6),
    ]


def _run(api, job, stream, calls, out, errs):
    name, make, n, rate, chan = job
    try:
        h = make()
        x = api.DeviceArray(2 * n)
        x.fill_synth(synth.SEED, channel=chan, stream=stream)
        cap = int(n / rate) + 64
        y = api.DeviceArray(2 * cap)
        y.zero(stream=stream)
        k = n
        for _ in range(calls):              # every call continues the stream: carried history / time state
            if name.startswith("fir"):
                h.process_stream(x, y, n, stream=stream)
            else:
                k = h.process_stream(x, n, y, cap, rate, stream=stream)
        m = min(1 << 20, 2 * k)
        out[name] = (k, y.to_numpy(m, stream=stream), y.to_numpy(m, offset=2 * k - m, stream=stream))
        h.close()
        x.free()
        y.free()
    except Exception as e:                  # noqa: BLE001 -- reported by the main thread
        errs.append((name, repr(e)))


def test_handles_driven_from_concurrent_host_threads_equal_the_same_handles_alone():
    import torch
    from simplefe_amd import api, lib as L
    LL = L.load()
    jobs = _jobs(api, L)
    alone, errs = {}, []
    for j in jobs:
        _run(api, j, None, 3, alone, errs)
    assert not errs, errs
    LL.sfe_dsp_fir_forget_calibrations()
    together = {}
    streams = [torch.cuda.Stream() for _ in jobs]
    threads = [threading.Thread(target=_run, args=(api, j, s.cuda_stream, 3, together, errs)) for j, s in zip(jobs, streams)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a thread did not finish"
    assert not errs, errs
    for name, _, _, _, _ in jobs:
        ka, ha, ta = alone[name]
        kt, ht, tt = together[name]
        assert ka == kt, (name, ka, kt)
        assert np.array_equal(ha, ht) and np.array_equal(ta, tt), name
