"""config::Qlpc::use_direct_mse / mae_optimization_steps on the GPU: the reference's experimental
estimators -- covariance-method LPC (src/lpc.rs:853-903) and its IRLS re-weighting (:814-850) -- selected in
perform_qlpc (src/coding.rs:333-351).

The GPU must equal the oracle's restatement bit for bit: R[], the unquantised solution of the Cholesky solve, the
quantised predictor, residual, Rice partition and bit counts, for every block shape and through every entry point.
What the oracle itself is pinned by is the reference's own tests of this path (tests/test_oracle_kat.py); the
solver is nalgebra's, outside the reference tree: beyond those tests this row is parity-unpinned."""
import numpy as np
import pytest

import util
from flacenc_rs_amd import _capi
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def handle():
    h = _capi.Handle(0)
    yield h
    h.close()


def gcfg(order, steps=0, **kw):
    return _capi.make_config(lpc_order=order, use_direct_mse=True, mae_optimization_steps=steps, **kw)


def ocfg(order, steps=0, **kw):
    return orc.make_config(lpc_order=order, use_direct_mse=True, mae_optimization_steps=steps, **kw)


def records_equal(g, o):
    for f in ("order", "shift", "precision", "rice_order", "status", "code_bits", "subframe_bits", "sum_quotients"):
        assert np.array_equal(g[f], o[f]), (f, g[f][:8], o[f][:8])
    assert np.array_equal(g["coefs"], o["coefs"])
    assert np.array_equal(g["rice_params"], o["rice_params"])


def exact(handle, x, bps, order, steps=0, **kw):
    x = np.ascontiguousarray(x, np.int32)
    gp, gres, gR, gA = handle.qlpc_batch(x, bps, gcfg(order, steps, **kw), want_fp=True)
    rp, rres, rR, rA = orc.qlpc_batch(x, bps, ocfg(order, steps, **kw))
    assert (gp["status"] == 0).all()
    assert np.array_equal(gR.view(np.uint64), rR.view(np.uint64)), "R[] bits"
    assert np.array_equal(gA.view(np.uint64), rA.view(np.uint64)), "solution bits"
    records_equal(gp, rp)
    assert np.array_equal(gres, rres)
    for k in range(x.shape[0]):
        o = int(gp["order"][k])
        assert np.array_equal(orc.decode_lpc(x[k][:o], gp["coefs"][k][:o], int(gp["shift"][k]), gres[k]), x[k])
    return gp


def batch(ns, n, bps, seed0):
    return np.stack([util.sine_noise(n, bps, 20 + 13 * (k % 17), 0.1 + 0.05 * (k % 9), 0.01 * (1 + k % 11),
                                     seed=seed0 + k, phase=0.1 * k) for k in range(ns)])


@pytest.mark.parametrize("ns,n,bps,order,window", [
    (9, 4096, 16, 8, "rectangle"),        # the reference's experimental preset: Rectangle window
    (9, 4096, 16, 10, ("tukey", 0.4)),
    (5, 4096, 16, 12, "rectangle"), (5, 4096, 16, 1, "rectangle"), (5, 4096, 24, 16, "rectangle"),
    (4, 8192, 24, 24, "rectangle"), (3, 8192, 24, 32, "rectangle"), (3, 16384, 24, 24, ("tukey", 0.1)),
    (2, 16384, 25, 32, "rectangle"),
])
def test_direct_mse_bit_exact(handle, ns, n, bps, order, window):
    exact(handle, batch(ns, n, bps, 300 + order), bps, order, window=window)


@pytest.mark.parametrize("n,order", [(4608, 12), (1152, 8), (576, 6), (100, 4), (64, 2), (20000, 16), (8191, 9),
                                      (128, 24), (77, 32)])
def test_direct_mse_ragged_blocks(handle, n, order):
    exact(handle, batch(4, n, 16, 7 * n), 16, order, window="rectangle")


def test_degenerate_blocks_take_the_regulariser_path(handle):
    """All-zero and constant blocks, a pure sine (rank-deficient Gram matrix): Cholesky fails, the diagonal is
    loaded with 1, 2, 4, ... until it succeeds (src/lpc.rs:887-896)."""
    n = 4096
    x = np.stack([np.zeros(n, np.int32), np.full(n, 1000, np.int32), util.quantize(util.sine(n, 64, 0.5), 16),
                  (np.arange(n) % 2 * 2000 - 1000).astype(np.int32), np.arange(n, dtype=np.int32) - 2048])
    exact(handle, x, 16, 8, window="rectangle")
    exact(handle, x, 16, 24, window="rectangle")


@pytest.mark.parametrize("n,bps,order,steps", [(4096, 16, 8, 1), (4096, 16, 10, 2), (1024, 16, 16, 4), (8192, 24, 12, 2),
                                                (16384, 16, 8, 1), (576, 16, 6, 3), (4096, 16, 24, 2)])
def test_irls_bit_exact(handle, n, bps, order, steps):
    """lpc_with_irls_mae: the weights go through f32::powf (restated glibc algorithm), the Gram sums through the
    f32 products w[t] * x[t]; every step's solve and the best-of choice must match."""
    x = batch(4, n, bps, 900 + order)
    x[1] = util.test_signal("sus109", 0)[:n] if n <= 8192 and bps == 16 else x[1]
    exact(handle, x, bps, order, steps, window="rectangle")


def test_irls_with_all_zero_block(handle):
    """normalizer 0 (lpc.rs:827): the weight is powf(inf, -1.2) = 0"""
    x = np.zeros((2, 1024), np.int32)
    x[1, 100] = 5
    exact(handle, x, 16, 8, 2, window="rectangle")


@pytest.mark.parametrize("order", [8, 12, 24])
def test_stereo_candidates(handle, order):
    n = 4096
    l, r = batch(5, n, 16, 900 + order), batch(5, n, 16, 1900 + order)
    frames = np.stack([l, r], axis=1)
    gp, gres = handle.stereo_qlpc_batch(frames, 16, gcfg(order, window="rectangle"))
    for f in range(frames.shape[0]):
        m, s = orc.stereo_to_midside(l[f], r[f])
        x = np.stack([l[f], r[f], m, s])
        rp, rres, _, _ = orc.qlpc_batch(x, np.array([16, 16, 16, 17], np.uint8), ocfg(order, window="rectangle"))
        records_equal(gp[f], rp)
        assert np.array_equal(gres[f], rres)


@pytest.mark.parametrize("n,bps,order,steps,use_fixed", [(4096, 16, 8, 0, True), (4096, 16, 10, 1, True),
                                                          (1152, 16, 8, 0, False), (8192, 24, 24, 0, True)])
def test_frame_pipeline_and_bytes(handle, n, bps, order, steps, use_fixed):
    """encode_stereo_frames + pack_stereo_frames with the experimental estimator == the oracle's controller and
    bit writer with the same switches: decisions, records, residual rows, frame bytes."""
    F = 5
    frames = _capi.sigen_frames(F, 2, n, bps, 36.0, 0.4, 0.04, seed=77 + order, nthreads=1)
    cfg = _capi.make_frame_config(gcfg(order, steps, window="rectangle"), use_fixed=use_fixed)
    res, resid = handle.encode_stereo_frames(frames, bps, cfg)
    ofc = orc.make_frame_config(ocfg(order, steps, window="rectangle"), use_fixed=use_fixed,
                                fixed=orc.make_fixed_config(sum_mode=orc.SUMABS_CANONICAL))
    want, wres = orc.encode_stereo_frames_cfg(frames, bps, ofc)
    assert res["channel_assignment"].tolist() == want["channel_assignment"].tolist()
    assert res["kind"].tolist() == want["kind"].tolist() and res["bits"].tolist() == want["bits"].tolist()
    assert np.array_equal(resid, wres)
    assert res.tobytes() == want.tobytes()
    packed = handle.pack_stereo_frames(frames, res, resid, bps, 44100)
    for f in range(F):
        assert packed[f] == orc.write_stereo_frame(res[f], frames[f, 0], frames[f, 1], bps, 44100, f,
                                                   resid[f, 0], resid[f, 1])


def test_independent_channels(handle):
    n, order, bps, ch = 4096, 10, 16, 3
    x = _capi.sigen_frames(3, ch, n, bps, 50.0, 0.3, 0.05, seed=4242, nthreads=1)
    cfg = _capi.make_frame_config(gcfg(order, window="rectangle"), use_fixed=True)
    res, resid = handle.encode_frames(x, bps, cfg)
    ofc = orc.make_frame_config(ocfg(order, window="rectangle"), use_fixed=True,
                                fixed=orc.make_fixed_config(sum_mode=orc.SUMABS_CANONICAL))
    for f in range(x.shape[0]):
        for c in range(ch):
            w = orc.encode_subframe(x[f, c], bps, ofc)
            g = res[f, c]
            assert int(g["kind"]) == w["kind"] and int(g["bits"]) == w["bits"], (f, c)
            if w["kind"] >= 2:
                assert np.array_equal(resid[f, c], w["residual"]), (f, c)


def test_the_reference_s_qualitative_checks_hold_on_the_gpu(handle):
    """src/lpc.rs:1297-1337 on the GPU's own numbers: on a 128-sample block of sus109 the covariance method
    predicts better than the windowed autocorrelation method."""
    sg = util.test_signal("sus109", 0)[:128][None, :]
    _, _, _, ga = handle.qlpc_batch(sg, 16, _capi.make_config(lpc_order=24, window=("tukey", 0.1)), want_fp=True)
    _, _, _, gd = handle.qlpc_batch(sg, 16, gcfg(24, window="rectangle"), want_fp=True)
    energy = (sg[0].astype(np.float64) ** 2).sum()
    ea = orc.compute_raw_errors(sg[0], ga[0, :24])[24:].astype(np.float64)
    ed = orc.compute_raw_errors(sg[0], gd[0, :24])[24:].astype(np.float64)
    assert (ea ** 2).sum() > (ed ** 2).sum() and energy > (ed ** 2).sum()


def test_config_limits(handle):
    x = batch(2, 32767, 16, 1)
    # lpc_with_irls_mae takes any block up to 32767 samples (src/constant.rs:57): above 16384 the weights leave the LDS
    exact(handle, x, 16, 8, 2, window="rectangle")
    exact(handle, x[:, :20000], 16, 12, 1, window=("tukey", 0.3))
    exact(handle, x, 16, 8, window="rectangle")       # without IRLS the largest block fits
    with pytest.raises(_capi.FlacencHipError) as e:
        handle.qlpc_batch(x[:, :4096], 16, gcfg(8, 65))
    assert e.value.code == _capi.ERR_BAD_CONFIG
