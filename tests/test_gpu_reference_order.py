"""FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER: with the autocorrelation summed in the order of the reference's
stable build (one sequential mul_add chain per lag, src/lpc.rs:533-548) the GPU must equal the oracle in
its REFERENCE mode bit for bit -- R[], unquantised and quantised coefficients, shift, order, residual,
Rice partition, every bit count -- not merely within the tolerance the default (canonical-order) mode is
held to.  Covered: the five BASELINE shapes, the reference's real-audio fixtures (src/resource/*.bin),
the committed golden vectors (oracle output in reference order), ragged block sizes, unaligned rows, and
the stereo / frame-level entry points (fused wave kernel with its phase 1 skipped).  The flag also covers
the other order-sensitive sum of the path: find_sum_abs_f32 (src/arrayutils.rs:496-506) inside the ApproxEnt
order selector of fixed_lpc, one sequential f32 chain per estimator partition in the stable build."""
import os

import numpy as np
import pytest

import util
from flacenc_rs_amd import _capi
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

REF = _capi.FLAG_REFERENCE_SUM_ORDER


@pytest.fixture(scope="module")
def handle():
    h = _capi.Handle(0)
    yield h
    h.close()


def gcfg(order, **kw):
    return _capi.make_config(lpc_order=order, flags=REF, **kw)


def ocfg(order, **kw):
    return orc.make_config(lpc_order=order, acorr=orc.ACORR_REFERENCE, **kw)


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
    assert np.array_equal(gR.view(np.uint64), rR.view(np.uint64)), "R[] bits differ from the reference order"
    assert np.array_equal(gA.view(np.uint64), rA.view(np.uint64)), "LPC coefficient bits"
    records_equal(gp, rp)
    assert np.array_equal(gres, rres)
    for k in range(x.shape[0]):
        o = int(gp["order"][k])
        assert np.array_equal(orc.decode_lpc(x[k][:o], gp["coefs"][k][:o], int(gp["shift"][k]), gres[k]), x[k])
    return gp


def batch(ns, n, bps, seed0):
    return np.stack([util.sine_noise(n, bps, 20 + 13 * (k % 17), 0.1 + 0.05 * (k % 9), 0.01 * (1 + k % 11),
                                     seed=seed0 + k, phase=0.1 * k) for k in range(ns)])


@pytest.mark.parametrize("ns,n,bps,order", [
    (70, 4096, 16, 8),      # BASELINE configs[1] (more than one 64-subframe wave, ragged last wave)
    (24, 4096, 16, 10),     # configs[0] / [3]: default order
    (6, 8192, 24, 24),      # configs[2]
    (6, 8192, 24, 32),      # configs[2], order-32 extension
    (4, 16384, 24, 24),     # configs[4]
    (3, 16384, 25, 32),
    (9, 4096, 16, 12),
    (5, 4096, 16, 1),
])
def test_baseline_shapes_bit_exact_in_reference_order(handle, ns, n, bps, order):
    exact(handle, batch(ns, n, bps, 100 + order), bps, order)


@pytest.mark.parametrize("n,order", [(4608, 12), (1152, 8), (576, 6), (100, 4), (64, 2), (20000, 16), (8191, 9)])
def test_ragged_blocks(handle, n, order):
    exact(handle, batch(5, n, 16, 7 * n), 16, order)


@pytest.mark.parametrize("window", ["rectangle", ("tukey", 0.0), ("tukey", 1.0), ("tukey", 0.1)])
def test_windows(handle, window):
    exact(handle, batch(6, 4096, 16, 31), 16, 10, window=window)


@pytest.mark.parametrize("name", ["sus109", "sus6", "ras22", "ras103"])
@pytest.mark.parametrize("ch", [0, 1])
def test_real_audio_fixtures(handle, name, ch):
    """The reference's own test signals (src/resource, test_helper.rs:81-125): 8192 samples each, cut into
    4096-sample blocks (order 8 and 10), one 8192 block (order 24) and 1152-sample blocks (order 12)."""
    s = util.test_signal(name, ch)
    exact(handle, s.reshape(2, 4096), 16, 8)
    exact(handle, s.reshape(2, 4096), 16, 10)
    exact(handle, s.reshape(1, 8192), 16, 24)
    exact(handle, s[: 7 * 1152].reshape(7, 1152), 16, 12)


GOLD = np.load(os.path.join(util.GOLDEN, "qlpc_golden.npz"))


@pytest.mark.parametrize("name", sorted({k.split("/")[0] for k in GOLD.files}))
def test_golden_vectors_exact(handle, name):
    """tests/golden/qlpc_golden.npz holds oracle output in reference order: in this mode every stored
    number must come out identically, floating point included."""
    n, order, bps = (int(v) for v in GOLD[f"{name}/meta"])
    x = GOLD[f"{name}/input"].astype(np.int32)[None, :]
    gp, gres, gR, gA = handle.qlpc_batch(x, bps, gcfg(order), want_fp=True)
    assert np.array_equal(gR[0, : order + 1].view(np.uint64), GOLD[f"{name}/autocorr"].view(np.uint64))
    assert np.array_equal(gA[0, :order].view(np.uint64), GOLD[f"{name}/lpc_coefs"].view(np.uint64))
    k = int(gp["order"][0])
    assert gp["coefs"][0][:k].tolist() == GOLD[f"{name}/coefs"].tolist()


def test_unaligned_rows_and_strides(handle):
    """Device-pointer entry with an odd row stride and a base pointer that is not 16-byte aligned: the tile
    loader's scalar path (and the generic kernels downstream)."""
    import torch
    ns, n, order = 9, 4096, 8
    x = batch(ns, n, 16, 555)
    stride = n + 3
    buf = torch.zeros(ns * stride + 5, dtype=torch.int32, device="cuda")
    view = buf[1:1 + ns * stride].view(ns, stride)
    view[:, :n] = torch.from_numpy(x).cuda()
    params = torch.zeros((ns, 352), dtype=torch.uint8, device="cuda")
    resid = torch.zeros((ns, n), dtype=torch.int32, device="cuda")
    R = torch.zeros((ns, 33), dtype=torch.float64, device="cuda")
    bps = torch.full((ns,), 16, dtype=torch.uint8, device="cuda")
    handle.qlpc_batch_device(gcfg(order), view.data_ptr(), ns, n, stride, bps.data_ptr(), params.data_ptr(),
                             resid.data_ptr(), n, autocorr_ptr=R.data_ptr(), sync=True)
    rp, rres, rR, _ = orc.qlpc_batch(x, 16, ocfg(order))
    assert np.array_equal(R.cpu().numpy().view(np.uint64), rR.view(np.uint64))
    gp = np.frombuffer(params.cpu().numpy().tobytes(), dtype=_capi.PARAMS_DTYPE)
    records_equal(gp, rp)
    assert np.array_equal(resid.cpu().numpy(), rres)


@pytest.mark.parametrize("order", [8, 10, 12, 16])
def test_stereo_candidates(handle, order):
    """flacenc_hip_stereo_qlpc_batch (L, R, M, S per frame): M and S are formed inside the tile loader."""
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
    (4096, 16, 8, False), (4096, 16, 8, True), (4096, 16, 10, True), (1152, 16, 8, True), (8192, 16, 24, False),
    # 24-bit material: the estimator's partition sums pass 2^24, where find_sum_abs_f32's sequential f32 chain
    # (arrayutils.rs:496-506) and an exactly rounded sum part ways -- fused kernel, big-block and generic paths
    (4096, 24, 8, True), (4096, 24, 12, True), (8192, 24, 24, True), (16384, 24, 24, True), (8192, 24, 8, True),
    (4608, 24, 10, True),
])
def test_frame_pipeline_and_bytes(handle, n, bps, order, use_fixed):
    """encode_stereo_frames + pack_stereo_frames in reference order == the oracle's encode_frame controller
    and bit writer in the reference's stable-build orders (autocorrelation: one chain per lag; ApproxEnt
    selector: one f32 chain per estimator partition): decisions, records, residual rows, frame bytes."""
    F = 6
    frames = _capi.sigen_frames(F, 2, n, bps, 36.0, 0.4, 0.04, seed=77 + order, nthreads=1)
    if bps == 24:
        frames[1] = _capi.sigen_frames(1, 2, n, bps, 300.0, 0.8, 0.001, seed=5, nthreads=1)[0]   # FixedLpc territory
        frames[2, 1] = frames[2, 0] // 3
    cfg = _capi.make_frame_config(gcfg(order), use_fixed=use_fixed)
    res, resid = handle.encode_stereo_frames(frames, bps, cfg)
    ofc = orc.make_frame_config(ocfg(order), use_fixed=use_fixed,
                                fixed=orc.make_fixed_config(sum_mode=orc.SUMABS_STABLE))
    want, wres = orc.encode_stereo_frames_cfg(frames, bps, ofc)
    assert res["channel_assignment"].tolist() == want["channel_assignment"].tolist()
    assert res["kind"].tolist() == want["kind"].tolist() and res["bits"].tolist() == want["bits"].tolist()
    assert np.array_equal(resid, wres)
    assert res.tobytes() == want.tobytes()
    packed = handle.pack_stereo_frames(frames, res, resid, bps, 44100)
    for f in range(F):
        assert packed[f] == orc.write_stereo_frame(res[f], frames[f, 0], frames[f, 1], bps, 44100, f,
                                                   resid[f, 0], resid[f, 1])


def _selector_corpus(n, bps):
    sigs = [util.sine_noise(n, bps, 200, 0.4, 0.05, seed=1), util.sine_noise(n, bps, 31, 0.7, 0.3, seed=2),
            util.quantize(util.sine(n, 100, 0.6), bps), (np.arange(n) // 7).astype(np.int32),
            ((np.arange(n) - n // 2) ** 2 // 400 % (1 << (bps - 2))).astype(np.int32),
            np.full(n, 77, np.int32), np.zeros(n, np.int32),
            util.quantize(util.noise(5, n, 0.999), bps),
            np.where(np.arange(n) % 2 == 0, 2 ** (bps - 1) - 1, -2 ** (bps - 1)).astype(np.int32),
            util.sine_noise(n, bps, 57, 0.9, 0.01, seed=3), util.sine_noise(n, bps, 1000, 0.5, 0.1, seed=4)]
    return np.stack(sigs).astype(np.int32)


@pytest.mark.parametrize("n,bps,parts,max_order", [
    (4096, 24, 16, 4), (8192, 24, 16, 4), (16384, 24, 16, 4), (16384, 25, 4, 4), (8192, 24, 64, 4),
    (8192, 24, 1, 4), (4096, 24, 8, 0), (4096, 16, 16, 4), (4096, 16, 2, 4),
    # generic kernel: ragged partitions (div_ceil sizes, a short or empty last partition, partitions shorter
    # than the warm-up), block sizes with no alignment at all
    (4608, 24, 16, 4), (1152, 24, 7, 4), (20000, 24, 33, 3), (100, 8, 64, 4), (8191, 24, 5, 4), (577, 24, 3, 2),
])
def test_fixed_selector_in_stable_sum_order(hooks_handle, n, bps, parts, max_order):
    """fixed_lpc with OrderSel::ApproxEnt in reference order: the selector's key of every order equals the
    oracle's with find_sum_abs_f32 as the stable build's sequential chain, and so do the chosen order, the Rice
    partition, the bit counts and the error signal.  The corpus is checked to be discriminating: at 24 bits
    some keys differ between the sequential chain and the exactly rounded sum."""
    handle = hooks_handle  # (debug_set_fixed_keys: the hooks build, same kernels)
    import torch
    x = _selector_corpus(n, bps)
    separating = bps == 24 and n in (8192, 16384) and parts == 16
    if separating:  # (the oracle's two modes differ on about one in ten of these)
        x = np.concatenate([x, batch(40, n, bps, 4000)])
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
    fc = orc.make_fixed_config(max_order=max_order, partitions=parts, sum_mode=orc.SUMABS_STABLE)
    fc_canon = orc.make_fixed_config(max_order=max_order, partitions=parts, sum_mode=orc.SUMABS_CANONICAL)
    differ = 0
    for k in range(ns):
        w = orc.fixed_lpc(x[k], bps, 2 ** 63, fc)
        assert ka[k, : max_order + 1].tolist() == w["estimate"][: max_order + 1], (k, "selector keys")
        differ += w["estimate"] != orc.fixed_lpc(x[k], bps, 2 ** 63, fc_canon)["estimate"]
        p = params[k]
        assert int(p["order"]) == w["order"] and int(keys[k]) == w["estimate"][w["order"]], k
        for fld in ("rice_order", "code_bits", "subframe_bits", "sum_quotients"):
            assert int(p[fld]) == int(w[fld]), (k, fld)
        assert p["rice_params"][: 1 << w["rice_order"]].tolist() == w["rice_params"].tolist()
        assert np.array_equal(resid[k], w["residual"]), k
    if separating:
        assert differ > 0, "corpus does not separate the two summation orders"


@pytest.mark.parametrize("n,parts", [(4096, 16), (8192, 16), (4608, 16), (4096, 32), (16384, 8)])
def test_fixed_selector_stereo_roles_in_stable_sum_order(handle, n, parts):
    """The same for the L, R, M, S roles of stereo frames (M and S formed inside the summing kernel)."""
    bps = 24
    x = _capi.sigen_frames(5, 2, n, bps, 90.0, 0.5, 0.02, seed=31, nthreads=1)
    x[1, 1] = x[1, 0] // 2 + 3
    x[2] = _capi.sigen_frames(1, 2, n, bps, 400.0, 0.9, 0.0005, seed=32, nthreads=1)[0]
    cfg = _capi.make_frame_config(gcfg(8), use_fixed=True, fixed_order_sel=1, fixed_partitions=parts)
    params, resid, keys = handle.fixed_lpc_batch(x, bps, cfg, stereo=True)
    fc = orc.make_fixed_config(partitions=parts, sum_mode=orc.SUMABS_STABLE)
    for f in range(x.shape[0]):
        l, r = x[f, 0], x[f, 1]
        for role, sig in enumerate([l, r, *orc.stereo_to_midside(l, r)]):
            w = orc.fixed_lpc(sig, bps + (1 if role == 3 else 0), 2 ** 63, fc)
            p = params[f, role]
            assert int(p["order"]) == w["order"] and int(keys[f, role]) == w["estimate"][w["order"]], (f, role)
            assert int(p["subframe_bits"]) == w["subframe_bits"] and int(p["code_bits"]) == w["code_bits"]
            assert np.array_equal(resid[f, role], w["residual"]), (f, role)


@pytest.mark.parametrize("channels,n,order", [(8, 4096, 10), (3, 4096, 8), (2, 8192, 12), (1, 1152, 8)])
def test_independent_channels_default_candidates_in_stable_sum_order(handle, channels, n, order):
    """flacenc_hip_encode_frames with the reference's default candidate set (use_fixed) on 24-bit material in
    reference order: kind, bits, record and residual of every channel == encode_subframe (coding.rs:384-418) with
    both reference summation orders."""
    bps = 24
    x = _capi.sigen_frames(4, channels, n, bps, 150.0, 0.5, 0.04, seed=channels * 1000 + n, nthreads=1)
    x[1, 0] = _capi.sigen_frames(1, 1, n, bps, 500.0, 0.9, 0.0003, seed=9, nthreads=1)[0, 0]
    x[2, channels - 1] = -5
    cfg = _capi.make_frame_config(gcfg(order), use_fixed=True)
    res, resid = handle.encode_frames(x, bps, cfg)
    ofc = orc.make_frame_config(ocfg(order), use_fixed=True, fixed=orc.make_fixed_config(sum_mode=orc.SUMABS_STABLE))
    for f in range(x.shape[0]):
        for c in range(channels):
            w = orc.encode_subframe(x[f, c], bps, ofc)
            g = res[f, c]
            assert int(g["kind"]) == w["kind"] and int(g["bits"]) == w["bits"], (f, c)
            if w["kind"] >= 2:
                assert np.array_equal(resid[f, c], w["residual"]), (f, c)
                src = w["lpc"] if w["kind"] == 3 else w["fixed"]
                assert int(g["params"]["subframe_bits"]) == int(src.subframe_bits)


def test_independent_channels(handle):
    """flacenc_hip_encode_frames (BASELINE configs[3]: 8 channels) in reference order."""
    n, order, bps, ch = 4096, 10, 16, 8
    frames = _capi.sigen_frames(3, ch, n, bps, 50.0, 0.3, 0.05, seed=4242, nthreads=1)
    cfg = _capi.make_frame_config(gcfg(order), use_fixed=False)
    res, resid = handle.encode_frames(frames, bps, cfg)
    flat = frames.reshape(-1, n)
    rp, rres, _, _ = orc.qlpc_batch(flat, bps, ocfg(order))
    lpc = res["kind"].reshape(-1) == 3
    assert lpc.any()
    got = res["params"].reshape(-1)
    records_equal(got[lpc], rp[lpc])
    assert np.array_equal(resid.reshape(-1, n)[lpc], rres[lpc])


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
    whole frame decision == the oracle with find_sum_abs_f32 in the stable order, and with few, large estimator
    partitions the corpus separates that order from the exactly rounded sum."""
    frames = _loud_16bit_frames(n)
    cfg = _capi.make_frame_config(gcfg(order), use_fixed=True, fixed_order_sel=1, fixed_partitions=parts)
    res, resid = handle.encode_stereo_frames(frames, 16, cfg)
    ofc = orc.make_frame_config(ocfg(order), use_fixed=True,
                                fixed=orc.make_fixed_config(partitions=parts, sum_mode=orc.SUMABS_STABLE))
    want, wres = orc.encode_stereo_frames_cfg(frames, 16, ofc)
    assert res["channel_assignment"].tolist() == want["channel_assignment"].tolist()
    assert res["kind"].tolist() == want["kind"].tolist() and res["bits"].tolist() == want["bits"].tolist()
    assert np.array_equal(resid, wres)
    assert res.tobytes() == want.tobytes()
    if (n, parts) in {(4096, 1), (4096, 2), (4608, 1), (4608, 2)}:
        fc = orc.make_fixed_config(partitions=parts, sum_mode=orc.SUMABS_STABLE)
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
