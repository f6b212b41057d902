"""Channel sharding behind the C ABI (VERDICT r3 missing 2 / next-round item 5): a group of channel
blocks, one per named device, equals ONE handle over all channels bit for bit.  The reference's
counterpart is one object per stream (libdsp/blkconv.h:35-62, libdsp/resample.h:33-61).  The GPU
box has one device, so the devices are {0, 0} and {0} x 8: the partition, the per-block streams,
the launch-all-then-wait order and the block handles are exactly what N devices would run."""
import numpy as np
import pytest

from simplefe_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from simplefe_amd import api as a
    return a


@pytest.fixture(scope="module")
def L():
    from simplefe_amd import lib
    return lib


def _streams(api, nch, n, first_seed=0):
    x = api.DeviceArray(2 * n * nch)
    for c in range(nch):
        x.fill_synth(synth.SEED, channel=first_seed + c, n_floats=2 * n, offset=2 * n * c)
    api.sync()
    return x


@pytest.mark.parametrize("devices", [[0, 0], [0] * 8, [0, 0, 0]])
@pytest.mark.parametrize("per_channel", [False, True])
def test_fir_group_equals_one_handle_over_all_channels(api, L, devices, per_channel):
    nch, n, calls = 64, 3840 * 9 + 124, 3          # even: channel rows stay 16-byte aligned for synth_fill
    taps = synth.taps_per_channel(nch) if per_channel else synth.taps_cfg2()
    one = api.Fir(taps, per_channel=True) if per_channel else api.Fir(taps, data_complex=True, n_channels=nch)
    grp = api.FirGroup(taps, nch, devices, per_channel=per_channel)
    sh = grp.shards()
    from simplefe_amd import shard
    assert [(d, f, c) for d, f, c, _, _ in sh] == [(devices[k],) + shard.channel_block(nch, len(devices), k) for k in range(len(devices))]
    assert sum(c for _, _, c, _, _ in sh) == nch and len({s for *_, s in sh}) == len(devices)      # a stream per block
    y1 = api.DeviceArray(2 * n * nch)
    ys = [api.DeviceArray(2 * n * c) for _, _, c, _, _ in sh]
    for call in range(calls):                                  # carried state across calls, per block
        x = _streams(api, nch, n, first_seed=100 * call)
        one.process_stream(x, y1, n)
        grp.process_stream([x.ptr + 8 * n * f for _, f, _, _, _ in sh], ys, n)
        grp.sync()
        api.sync()
        want = y1.to_numpy().reshape(nch, 2 * n)
        for (_, f, c, _, _), y in zip(sh, ys):
            assert np.array_equal(y.to_numpy().reshape(c, 2 * n), want[f:f + c]), (call, f)
    grp.reset()
    one.reset()
    x = _streams(api, nch, n, first_seed=7)
    one.process_stream(x, y1, n)
    grp.process_stream([x.ptr + 8 * n * f for _, f, _, _, _ in sh], ys, n)
    grp.sync()
    api.sync()
    assert np.array_equal(ys[-1].to_numpy().reshape(-1, 2 * n), y1.to_numpy().reshape(nch, 2 * n)[sh[-1][1]:])
    grp.close()


@pytest.mark.parametrize("which,U,rate,devices", [("decimate", 1, 8.0, [0, 0]), ("resample", 3, 5.0 / 3.0, [0] * 4),
                                                  ("resample", 4, 1.77, [0, 0])])
def test_rs_group_equals_one_handle_over_all_channels(api, L, which, U, rate, devices):
    nch, n = 8, 4096 * 6
    taps = {1: synth.taps_cfg4(), 3: synth.taps_cfg3(), 4: synth.lowpass_taps(31, 0.18).astype(np.float32) * 4}[U]
    mode = L.RS_DECIMATE if which == "decimate" else L.RS_RESAMPLE
    rate = float(np.float32(rate))
    cap = int(n / rate) + 16
    one = api.Rs(taps, U, 4096, mode=mode, data_complex=True, n_channels=nch)
    grp = api.RsGroup(taps, U, 4096, nch, devices, mode=mode)
    sh = grp.shards()
    y1 = api.DeviceArray(2 * cap * nch)
    ys = [api.DeviceArray(2 * cap * c) for _, _, c, _, _ in sh]
    for call in range(3):
        x = _streams(api, nch, n, first_seed=10 * call)
        k1 = one.process_stream(x, n, y1, cap, rate)
        kg = grp.process_stream([x.ptr + 8 * n * f for _, f, _, _, _ in sh], n, ys, cap, rate)
        grp.sync()
        api.sync()
        assert k1 == kg
        want = y1.to_numpy().reshape(nch, 2 * cap)[:, : 2 * k1]
        for (_, f, c, _, _), y in zip(sh, ys):
            assert np.array_equal(y.to_numpy().reshape(c, 2 * cap)[:, : 2 * k1], want[f:f + c]), (call, f)
    grp.close()


def test_group_argument_checks(api, L):
    with pytest.raises(api.SfeError) as e:
        api.FirGroup(synth.taps_cfg2(), 2, [0, 0, 0])          # more devices than channels
    assert e.value.code == L.SFE_EINVAL
    with pytest.raises(api.SfeError) as e:
        api.FirGroup(synth.taps_cfg2(), 4, [0, 99])            # no such device
    assert e.value.code == L.SFE_ENODEV


def test_group_call_that_fails_part_way_is_refused_until_reset(api, L):
    """A group call that fails after some shards have taken their launch leaves the shards out of step (for the resamplers:
    their time states): further calls return SFE_ESTATE until _reset (include/sfe_dsp.h).  A failure at the FIRST shard has
    moved nothing and leaves the group usable."""
    from simplefe_amd.lib import SfeError
    nch, n = 4, 8192
    grp = api.FirGroup(synth.taps_cfg2(), nch, [0, 0])
    x = _streams(api, nch, n)
    ys = [api.DeviceArray(2 * n * 2), api.DeviceArray(2 * n * 2)]
    good_in = [x.ptr, x.ptr + 8 * n * 2]
    with pytest.raises(SfeError):                                   # shard 0 refuses: nothing has moved
        grp.process_stream([0, good_in[1]], ys, n)
    grp.process_stream(good_in, ys, n)
    with pytest.raises(SfeError):                                   # shard 1 refuses after shard 0 has run
        grp.process_stream([good_in[0], 0], ys, n)
    with pytest.raises(SfeError) as e:
        grp.process_stream(good_in, ys, n)
    assert e.value.code == L.SFE_ESTATE and "out of step" in str(e.value)
    grp.reset()
    grp.process_stream(good_in, ys, n)
    grp.sync()
    grp.close()
