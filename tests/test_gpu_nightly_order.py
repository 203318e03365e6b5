"""FLACENC_HIP_FLAG_NIGHTLY_SUM_ORDER: the two order-sensitive sums of the path in the order of the reference's
`simd-nightly` build -- the build its published speed and compression figures come from
(report/report.nightly.md).

* autocorrelation: weighted_auto_correlation_simd (src/lpc.rs:510-531) over weighted_delay_prod_sum_impl
  (:439-500): per lag 8 or 16 strided f64 lane chains over the aligned body of the windowed buffer (a
  SimdVec<f32, 16>, 64-byte aligned, lpc.rs:710), scalar chains over head and foot, an ordered lane sum;
* find_sum_abs_f32 (src/arrayutils.rs:459-506) inside the ApproxEnt selector: 16 f32 lane chains + head / foot.

The GPU must equal the oracle's restatement of those orders (ACORR_NIGHTLY / SUMABS_NIGHTLY) bit for bit: R[],
coefficients, every integer output, frame bytes.  The order is defined up to lpc_order 15 (wider vectors split at
allocator-dependent addresses): above, the configuration is refused."""
import numpy as np
import pytest

import util
from flacenc_rs_amd import _capi
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

NIGHTLY = _capi.FLAG_NIGHTLY_SUM_ORDER


@pytest.fixture(scope="module")
def handle():
    h = _capi.Handle(0)
    yield h
    h.close()


def gcfg(order, **kw):
    return _capi.make_config(lpc_order=order, flags=NIGHTLY, **kw)


def ocfg(order, **kw):
    return orc.make_config(lpc_order=order, acorr=orc.ACORR_NIGHTLY, **kw)


def records_equal(g, o):
    for f in ("order", "shift", "precision", "rice_order", "status", "code_bits", "subframe_bits", "sum_quotients"):
        assert np.array_equal(g[f], o[f]), (f, g[f][:8], o[f][:8])
    assert np.array_equal(g["coefs"], o["coefs"])
    assert np.array_equal(g["rice_params"], o["rice_params"])


def exact(handle, x, bps, order, **kw):
    x = np.ascontiguousarray(x, np.int32)
    gp, gres, gR, gA = handle.qlpc_batch(x, bps, gcfg(order, **kw), want_fp=True)
    rp, rres, rR, rA = orc.qlpc_batch(x, bps, ocfg(order, **kw))
    assert (gp["status"] == 0).all()
    assert np.array_equal(gR.view(np.uint64), rR.view(np.uint64)), "R[] bits differ from the simd-nightly order"
    assert np.array_equal(gA.view(np.uint64), rA.view(np.uint64)), "LPC coefficient bits"
    records_equal(gp, rp)
    assert np.array_equal(gres, rres)
    # and the order is a different one: R[] is not the stable build's (else the test proves nothing)
    sR = orc.qlpc_batch(x, bps, orc.make_config(lpc_order=order, acorr=orc.ACORR_REFERENCE, **kw))[2]
    return int((gR.view(np.uint64) != sR.view(np.uint64)).sum())


def batch(ns, n, bps, seed0):
    return np.stack([util.sine_noise(n, bps, 20 + 13 * (k % 17), 0.1 + 0.05 * (k % 9), 0.01 * (1 + k % 11),
                                     seed=seed0 + k, phase=0.1 * k) for k in range(ns)])


@pytest.mark.parametrize("ns,n,bps,order", [
    (70, 4096, 16, 8),      # BASELINE configs[1]
    (24, 4096, 16, 10),     # configs[0] / [3]: the reference's default order
    (9, 4096, 16, 12), (9, 4096, 16, 15), (5, 4096, 16, 1), (5, 4096, 16, 7), (6, 4096, 24, 9),
    (6, 8192, 24, 12), (4, 16384, 24, 15), (3, 16384, 25, 8),
])
def test_shapes_bit_exact_in_nightly_order(handle, ns, n, bps, order):
    differs = exact(handle, batch(ns, n, bps, 100 + order), bps, order)
    assert differs > 0


@pytest.mark.parametrize("n,order", [(4608, 12), (1152, 8), (576, 6), (100, 4), (64, 2), (20000, 15), (8191, 9),
                                      (4097, 10), (77, 15)])
def test_ragged_blocks(handle, n, order):
    exact(handle, batch(5, n, 16, 7 * n), 16, order)


@pytest.mark.parametrize("window", ["rectangle", ("tukey", 0.0), ("tukey", 1.0), ("tukey", 0.1)])
def test_windows(handle, window):
    exact(handle, batch(6, 4096, 16, 31), 16, 10, window=window)


@pytest.mark.parametrize("name", ["sus109", "sus6", "ras22", "ras103"])
@pytest.mark.parametrize("ch", [0, 1])
def test_real_audio_fixtures(handle, name, ch):
    """The reference's own test signals (src/resource, test_helper.rs:81-125)."""
    s = util.test_signal(name, ch)
    exact(handle, s.reshape(2, 4096), 16, 8)
    exact(handle, s.reshape(2, 4096), 16, 10)
    exact(handle, s.reshape(1, 8192), 16, 14)
    exact(handle, s[: 7 * 1152].reshape(7, 1152), 16, 12)


@pytest.mark.parametrize("order", [8, 10, 12, 15])
def test_stereo_candidates(handle, order):
    n = 4096
    l, r = batch(7, n, 16, 900 + order), batch(7, n, 16, 1900 + order)
    frames = np.stack([l, r], axis=1)
    gp, gres = handle.stereo_qlpc_batch(frames, 16, gcfg(order))
    for f in range(frames.shape[0]):
        m, s = orc.stereo_to_midside(l[f], r[f])
        x = np.stack([l[f], r[f], m, s])
        rp, rres, _, _ = orc.qlpc_batch(x, np.array([16, 16, 16, 17], np.uint8), ocfg(order))
        records_equal(gp[f], rp)
        assert np.array_equal(gres[f], rres)


@pytest.mark.parametrize("n,bps,order,use_fixed", [
    (4096, 16, 8, False), (4096, 16, 10, True), (1152, 16, 8, True), (4096, 24, 8, True), (4096, 24, 12, True),
    (8192, 24, 15, True), (16384, 24, 10, True), (4608, 24, 10, True),
])
def test_frame_pipeline_and_bytes(handle, n, bps, order, use_fixed):
    """encode_stereo_frames + pack_stereo_frames == the oracle's encode_frame controller and bit writer with
    both sums in the simd-nightly order: decisions, records, residual rows, frame bytes."""
    F = 6
    frames = _capi.sigen_frames(F, 2, n, bps, 36.0, 0.4, 0.04, seed=77 + order, nthreads=1)
    if bps == 24:
        frames[1] = _capi.sigen_frames(1, 2, n, bps, 300.0, 0.8, 0.001, seed=5, nthreads=1)[0]
        frames[2, 1] = frames[2, 0] // 3
    cfg = _capi.make_frame_config(gcfg(order), use_fixed=use_fixed)
    res, resid = handle.encode_stereo_frames(frames, bps, cfg)
    ofc = orc.make_frame_config(ocfg(order), use_fixed=use_fixed,
                                fixed=orc.make_fixed_config(sum_mode=orc.SUMABS_NIGHTLY))
    want, wres = orc.encode_stereo_frames_cfg(frames, bps, ofc)
    assert res["channel_assignment"].tolist() == want["channel_assignment"].tolist()
    assert res["kind"].tolist() == want["kind"].tolist() and res["bits"].tolist() == want["bits"].tolist()
    assert np.array_equal(resid, wres)
    assert res.tobytes() == want.tobytes()
    packed = handle.pack_stereo_frames(frames, res, resid, bps, 44100)
    for f in range(F):
        assert packed[f] == orc.write_stereo_frame(res[f], frames[f, 0], frames[f, 1], bps, 44100, f,
                                                   resid[f, 0], resid[f, 1])


@pytest.mark.parametrize("n,bps,parts,max_order", [
    (4096, 24, 16, 4), (8192, 24, 16, 4), (16384, 24, 16, 4), (16384, 25, 4, 4), (8192, 24, 64, 4), (8192, 24, 1, 4),
    (4096, 16, 16, 4), (4608, 24, 16, 4), (1152, 24, 7, 4), (20000, 24, 33, 3), (100, 8, 64, 4), (8191, 24, 5, 4),
    (577, 24, 3, 2), (4096, 24, 12, 4),
])
def test_fixed_selector_in_nightly_sum_order(hooks_handle, n, bps, parts, max_order):
    """fixed_lpc's ApproxEnt keys with find_sum_abs_f32 in the simd-nightly order (partition p of a 64-byte aligned
    SimdVec<i32, 16> starts at element offset p * partition_size: head up to the next multiple of 16, 16 lane
    chains over the body, foot), for every order; chosen order, Rice partition, bit counts, error signal."""
    handle = hooks_handle  # (debug_set_fixed_keys: the hooks build, same kernels)
    import torch
    x = np.concatenate([batch(12, n, bps, 4000),
                        np.stack([util.quantize(util.noise(5, n, 0.999), bps), (np.arange(n) // 7).astype(np.int32),
                                  np.zeros(n, np.int32)])])
    ns = x.shape[0]
    bpsv = np.full(ns, bps, np.uint8)
    cfg = _capi.make_frame_config(gcfg(8), use_fixed=True, fixed_order_sel=1, fixed_partitions=parts,
                                  fixed_max_order=max_order)
    keys_all = torch.zeros((ns, 8), dtype=torch.int64, device="cuda")
    handle.debug_set_fixed_keys(keys_all.data_ptr())
    try:
        params, resid, keys = handle.fixed_lpc_batch(x, bpsv, cfg)
    finally:
        handle.debug_set_fixed_keys(0)
    ka = keys_all.cpu().numpy().astype(np.uint64)
    fc = orc.make_fixed_config(max_order=max_order, partitions=parts, sum_mode=orc.SUMABS_NIGHTLY)
    for k in range(ns):
        w = orc.fixed_lpc(x[k], bps, 2 ** 63, fc)
        assert ka[k, : max_order + 1].tolist() == w["estimate"][: max_order + 1], (k, "selector keys")
        p = params[k]
        assert int(p["order"]) == w["order"] and int(keys[k]) == w["estimate"][w["order"]], k
        for fld in ("rice_order", "code_bits", "subframe_bits", "sum_quotients"):
            assert int(p[fld]) == int(w[fld]), (k, fld)
        assert np.array_equal(resid[k], w["residual"]), k


def test_independent_channels(handle):
    """flacenc_hip_encode_frames (BASELINE configs[3]: 8 channels, default candidates) in the nightly order."""
    n, order, bps, ch = 4096, 10, 24, 8
    x = _capi.sigen_frames(3, ch, n, bps, 50.0, 0.3, 0.05, seed=4242, nthreads=1)
    cfg = _capi.make_frame_config(gcfg(order), use_fixed=True)
    res, resid = handle.encode_frames(x, bps, cfg)
    ofc = orc.make_frame_config(ocfg(order), use_fixed=True, fixed=orc.make_fixed_config(sum_mode=orc.SUMABS_NIGHTLY))
    for f in range(x.shape[0]):
        for c in range(ch):
            w = orc.encode_subframe(x[f, c], bps, ofc)
            g = res[f, c]
            assert int(g["kind"]) == w["kind"] and int(g["bits"]) == w["bits"], (f, c)
            if w["kind"] >= 2:
                assert np.array_equal(resid[f, c], w["residual"]), (f, c)


def test_orders_beyond_15_and_mixed_flags_are_refused(handle):
    x = batch(2, 4096, 16, 1)
    with pytest.raises(_capi.FlacencHipError) as e:
        handle.qlpc_batch(x, 16, _capi.make_config(lpc_order=16, flags=NIGHTLY))
    assert e.value.code == _capi.ERR_UNSUPPORTED
    with pytest.raises(_capi.FlacencHipError) as e:
        handle.qlpc_batch(x, 16, _capi.make_config(lpc_order=8, flags=NIGHTLY | _capi.FLAG_REFERENCE_SUM_ORDER))
    assert e.value.code == _capi.ERR_BAD_CONFIG


def _loud_16bit_frames(n):
    """16-bit stereo frames whose fixed-LPC error sums pass 2^24 per estimator partition at the higher orders:
    full-scale alternation (+ noise, so that the f32 roundings depend on the order of the additions), opposite
    phases in the two channels (a 17-bit side channel), full-scale white noise, ordinary material, and 48 frames
    of alternation at random amplitudes."""
    rng = np.random.default_rng(20261003)
    alt = np.where(np.arange(n) % 2 == 0, 32767, -32768).astype(np.int64)
    jit = rng.integers(-1500, 1500, size=(8, n))
    f = np.zeros((54, 2, n), np.int64)
    f[0, 0], f[0, 1] = alt + jit[0], alt + jit[1]
    f[1, 0], f[1, 1] = alt + jit[2], -alt + jit[3]
    f[2, 0], f[2, 1] = rng.integers(-32768, 32768, size=n), rng.integers(-32768, 32768, size=n)
    f[3] = _capi.sigen_frames(1, 2, n, 16, 120.0, 0.5, 0.02, seed=8, nthreads=1)[0]
    f[4, 0], f[4, 1] = alt * (np.arange(n) < n // 2) + jit[4] // 8, f[3, 0]   # loud first half only
    f[5, 0], f[5, 1] = (alt // 2) + jit[5], rng.integers(-30000, 30000, size=n)
    rng = np.random.default_rng(7)
    sgn = np.where(np.arange(n) % 2 == 0, 1, -1).astype(np.int64)
    for i in range(48):
        for c in range(2):
            amp, j = rng.integers(20000, 32768), rng.integers(100, 4000)
            f[6 + i, c] = sgn * amp * (1 if c == 0 or i % 2 else -1) + rng.integers(-j, j, size=n)
    return np.clip(f, -32768, 32767).astype(np.int32)


@pytest.mark.parametrize("n,order,parts", [(4096, 8, 16), (4096, 10, 2), (4096, 12, 1), (4096, 8, 64), (4608, 10, 16),
                                           (4608, 8, 1), (4608, 12, 2)])
def test_sixteen_bit_partitions_past_2_pow_24(handle, n, order, parts):
    """On material of at most 16 bits the fused kernel runs without sumabs_reference_kernel in front of it: its
    exact sums of |e| are the reference's f32 chains while a partition stays below 2^24, and it walks the
    partitions that do not itself.  Here most partitions of the loud frames are past 2^24 at orders 2..4: the
    whole frame decision == the oracle with find_sum_abs_f32 in the nightly order, and with few, large estimator
    partitions the corpus separates that order from the exactly rounded sum."""
    frames = _loud_16bit_frames(n)
    cfg = _capi.make_frame_config(gcfg(order), use_fixed=True, fixed_order_sel=1, fixed_partitions=parts)
    res, resid = handle.encode_stereo_frames(frames, 16, cfg)
    ofc = orc.make_frame_config(ocfg(order), use_fixed=True,
                                fixed=orc.make_fixed_config(partitions=parts, sum_mode=orc.SUMABS_NIGHTLY))
    want, wres = orc.encode_stereo_frames_cfg(frames, 16, ofc)
    assert res["channel_assignment"].tolist() == want["channel_assignment"].tolist()
    assert res["kind"].tolist() == want["kind"].tolist() and res["bits"].tolist() == want["bits"].tolist()
    assert np.array_equal(resid, wres)
    assert res.tobytes() == want.tobytes()
    if (n, parts) in {(4096, 2), (4608, 1)}:
        fc = orc.make_fixed_config(partitions=parts, sum_mode=orc.SUMABS_NIGHTLY)
        fc_canon = orc.make_fixed_config(partitions=parts, sum_mode=orc.SUMABS_CANONICAL)
        differ = 0
        for f in range(frames.shape[0]):
            l, r = frames[f, 0], frames[f, 1]
            for role, sig in enumerate([l, r, *orc.stereo_to_midside(l, r)]):
                b = 16 + (role == 3)
                differ += orc.fixed_lpc(sig, b, 2 ** 63, fc)["estimate"] != orc.fixed_lpc(sig, b, 2 ** 63, fc_canon)["estimate"]
        assert differ > 0, "corpus does not separate the summation orders"
    # independent channels (encode_subframe per channel) through the same kernel family
    ch = frames[:6].reshape(3, 4, n)
    cres, cresid = handle.encode_frames(ch, 16, cfg)
    for f in range(3):
        for c in range(4):
            w = orc.encode_subframe(ch[f, c], 16, ofc)
            g = cres[f, c]
            assert int(g["kind"]) == w["kind"] and int(g["bits"]) == w["bits"], (f, c)
            if w["kind"] >= 2:
                assert np.array_equal(cresid[f, c], w["residual"]), (f, c)
