"""Blocks of 8192 / 16384 samples at LPC order 13..32 (BASELINE configs[2] and configs[4]) run on the
pass-structured kernels of csrc/qlpc_bigblock.cpp: everything must equal the oracle in canonical order bit
for bit (R[], coefficients, records, residual), the kernels the shapes ran on before (FLAG_GENERIC_KERNEL)
byte for byte, and decode back to the input -- including what only these shapes can reach: Rice partition
orders 7 and 8 winning or losing against finer ones, a warm-up of up to 32 samples inside the first
partition, residuals of 2^26 and more (handed back to the generic kernel's literal bit tables), the
reference summation order, and a plain batch whose size is not a multiple of the workgroup's four rows."""
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


def records_equal(g, o, what=""):
    for f in ("order", "shift", "precision", "rice_order", "status", "code_bits", "subframe_bits", "sum_quotients"):
        assert np.array_equal(g[f], o[f]), (what, f, g[f][:8], o[f][:8])
    assert np.array_equal(g["coefs"], o["coefs"]), what
    assert np.array_equal(g["rice_params"], o["rice_params"]), what


def check(handle, x, bps, order, flags=0, acorr=orc.ACORR_CANONICAL, **kw):
    x = np.ascontiguousarray(x, np.int32)
    cfg = _capi.make_config(lpc_order=order, flags=flags, **kw)
    gp, gres, gR, gA = handle.qlpc_batch(x, bps, cfg, want_fp=True)
    op, ores, oR, oA = orc.qlpc_batch(x, bps, orc.make_config(lpc_order=order, acorr=acorr, **kw))
    assert np.array_equal(gR.view(np.uint64), oR.view(np.uint64)), "R[] bits"
    assert np.array_equal(gA.view(np.uint64), oA.view(np.uint64)), "LPC coefficient bits"
    records_equal(gp, op, "oracle")
    assert np.array_equal(gres, ores)
    # the generic kernel (what these shapes ran on before) must agree byte for byte
    pp, pres, _, _ = handle.qlpc_batch(x, bps, _capi.make_config(lpc_order=order, flags=flags | _capi.FLAG_GENERIC_KERNEL, **kw))
    assert gp.tobytes() == pp.tobytes() and np.array_equal(gres, pres)
    for k in range(x.shape[0]):
        o = int(gp["order"][k])
        assert np.array_equal(orc.decode_lpc(x[k][:o], gp["coefs"][k][:o], int(gp["shift"][k]), gres[k]), x[k])
    return gp


def batch(ns, n, bps, seed0, namp=0.01):
    return np.stack([util.sine_noise(n, bps, 20 + 13 * (k % 17), 0.1 + 0.05 * (k % 9), namp * (1 + k % 11),
                                     seed=seed0 + k, phase=0.1 * k) for k in range(ns)])


@pytest.mark.parametrize("n", [8192, 16384])
# (orders up to 12 on these blocks took the generic kernel until round 3: 9- and 13-lag autocorrelation instances,
# residual buckets 8 and 12)
@pytest.mark.parametrize("order", [1, 2, 5, 8, 9, 10, 12, 13, 16, 17, 24, 25, 32])
@pytest.mark.parametrize("bps", [16, 24])
def test_plain_batches(handle, n, order, bps):
    ns = 7 if n == 8192 else 5   # not a multiple of 4: the last workgroup has idle rows
    gp = check(handle, batch(ns, n, bps, 10 * order + bps), bps, order)
    assert (gp["status"] == 0).all()


@pytest.mark.parametrize("n,order", [(8192, 24), (8192, 32), (16384, 24), (16384, 32), (16384, 13), (8192, 8), (8192, 10),
                                     (16384, 12), (16384, 3)])
def test_stereo_candidates(handle, n, order):
    bps = 24
    l, r = batch(3, n, bps, 100 + order), batch(3, n, bps, 700 + order)
    frames = np.stack([l, r], axis=1)
    gp, gres = handle.stereo_qlpc_batch(frames, bps, _capi.make_config(lpc_order=order))
    pp, pres = handle.stereo_qlpc_batch(frames, bps, _capi.make_config(lpc_order=order, flags=_capi.FLAG_GENERIC_KERNEL))
    assert gp.tobytes() == pp.tobytes() and np.array_equal(gres, pres)
    for f in range(frames.shape[0]):
        m, s = orc.stereo_to_midside(l[f], r[f])
        x = np.stack([l[f], r[f], m, s])
        op, ores, _, _ = orc.qlpc_batch(x, np.array([bps, bps, bps, bps + 1], np.uint8),
                                        orc.make_config(lpc_order=order, acorr=orc.ACORR_CANONICAL))
        records_equal(gp[f], op)
        assert np.array_equal(gres[f], ores)


def test_coarse_partition_orders_win_and_lose(handle):
    """Signals whose statistics change at pass granularity (one 4096-sample pass loud, the next quiet) and
    stationary ones: the merged-pass orders (partition order 0 / 1 for 8192, 0..2 for 16384) must be chosen
    exactly when the reference chooses them."""
    rng = np.random.default_rng(3)
    for n in (8192, 16384):
        rows = []
        for k in range(6):
            amp = np.repeat(rng.choice([3.0, 40.0, 900.0, 20000.0], size=n // 4096), 4096) if k % 2 else np.full(n, 50.0 * (k + 1))
            rows.append(np.round(rng.normal(0, 1, n) * amp).astype(np.int32))
        gp = check(handle, np.stack(rows), 24, 24)
        print(n, "rice orders:", gp["rice_order"].tolist())
    flat = np.round(rng.normal(0, 300, (3, 16384))).astype(np.int32)
    gp = check(handle, flat, 24, 32)
    assert (gp["rice_order"] <= 2).any()


@pytest.mark.parametrize("max_p", [2, 14, 30])
def test_huge_residuals_go_through_the_literal_tables(handle, max_p):
    """25-bit full-scale noise: zig-zag residuals of 2^26 and more, where the reference's chunk-clamped
    wrapping sums (rice.rs:75-98) differ from exact ones -- the big-block kernel hands these subframes to the
    generic kernel's literal path; mixed with ordinary subframes in one batch."""
    x = np.stack([util.quantize(util.noise(1, 8192, 1.0), 25), batch(1, 8192, 25, 5)[0],
                  util.quantize(util.noise(2, 8192, 1.0), 25), batch(1, 8192, 25, 6)[0],
                  np.where(np.arange(8192) % 2 == 0, (1 << 24) - 1, -(1 << 24)).astype(np.int32)])
    check(handle, x, 25, 24, max_rice_parameter=max_p)
    y = np.stack([util.quantize(util.noise(9, 16384, 1.0), 25), batch(1, 16384, 25, 7)[0]])
    check(handle, y, 25, 32, max_rice_parameter=max_p)
    # the low order buckets' clean-up launches (qlpc_marked_kernel<8 / 10 / 12>, workgroups of 512 / 1024 threads)
    for order in (8, 10, 12, 3):
        check(handle, x, 25, order, max_rice_parameter=max_p)
    check(handle, y, 25, 9, max_rice_parameter=max_p)


def test_degenerate_signals(handle):
    n = 8192
    x = np.stack([np.zeros(n, np.int32), np.full(n, -7, np.int32), np.arange(n, dtype=np.int32) - 4000,
                  ((np.arange(n) % 2) * 2 - 1).astype(np.int32) * 8388607,
                  np.concatenate([np.zeros(n - 1, np.int32), [8388607]]).astype(np.int32)])
    check(handle, x, 24, 32)
    check(handle, x, 24, 16, window="rectangle")


@pytest.mark.parametrize("n,order", [(8192, 24), (16384, 32), (8192, 10), (16384, 8)])
def test_reference_summation_order(handle, n, order):
    check(handle, batch(4, n, 24, 77), 24, order, flags=_capi.FLAG_REFERENCE_SUM_ORDER, acorr=orc.ACORR_REFERENCE)


@pytest.mark.parametrize("n,order,stereo", [(8192, 24, True), (8192, 32, True), (16384, 24, True), (16384, 32, True),
                                            (8192, 16, False), (16384, 20, False), (4096, 24, True), (4096, 16, False)])
def test_config3_config5_unflagged_order_is_the_references(handle, n, order, stereo):
    """BASELINE configs[2] / [4] (8192 / 16384 samples -- and 4096 --, 24-bit, orders from 16): with flags = 0 the product sums as the
    reference's stable build does (weighted_auto_correlation_nosimd, src/lpc.rs:533-548) -- R[], the unquantised and
    the quantised coefficients, residuals and Rice partitions are bit-equal to the oracle's ACORR_REFERENCE mode."""
    x = batch(4 if stereo else 5, n, 24, 1234 + n + order)
    gcfg = _capi.make_config(lpc_order=order)  # flags = 0
    ocfg = orc.make_config(lpc_order=order, acorr=orc.ACORR_REFERENCE)
    if stereo:
        frames = x.reshape(-1, 2, n)
        gp, gres = handle.stereo_qlpc_batch(frames, 24, gcfg)
        for f in range(frames.shape[0]):
            m, s = orc.stereo_to_midside(frames[f, 0], frames[f, 1])
            op, ores, _, _ = orc.qlpc_batch(np.stack([frames[f, 0], frames[f, 1], m, s]), np.array([24, 24, 24, 25], np.uint8), ocfg)
            records_equal(gp[f], op)
            assert np.array_equal(gres[f], ores)
    gp, gres, gR, gA = handle.qlpc_batch(x, 24, gcfg, want_fp=True)
    op, ores, oR, oA = orc.qlpc_batch(x, 24, ocfg)
    assert np.array_equal(gR.view(np.uint64), oR.view(np.uint64)), "R[] is the reference's sequential chain"
    assert np.array_equal(gA.view(np.uint64), oA.view(np.uint64))
    records_equal(gp, op)
    assert np.array_equal(gres, ores)
    # unaligned rows and the generic kernel change the kernels, not the sums
    pp, pres, pR, _ = handle.qlpc_batch(x, 24, _capi.make_config(lpc_order=order, flags=_capi.FLAG_GENERIC_KERNEL), want_fp=True)
    assert np.array_equal(pR.view(np.uint64), oR.view(np.uint64)) and gp.tobytes() == pp.tobytes()
    # the oracle's "canonical" mode (what the product computes unflagged) is that same order on these shapes ...
    _, _, cR, _ = orc.qlpc_batch(x, 24, orc.make_config(lpc_order=order, acorr=orc.ACORR_CANONICAL))
    assert np.array_equal(cR.view(np.uint64), oR.view(np.uint64))
    # ... which is not the chunk tree (the corpus separates the two orders)
    w = orc.window_weights(("tukey", 0.4), n)
    tree = np.stack([orc.auto_correlation(order + 1, orc.fill_windowed_signal(x[i], w), canonical=True) for i in range(x.shape[0])])
    assert not np.array_equal(tree.view(np.uint64), oR[:, : order + 1].view(np.uint64))


@pytest.mark.parametrize("n,order", [(8192, 15), (16384, 12), (4096, 15), (4608, 24), (2048, 24), (1000, 8), (20000, 10)])
def test_shapes_next_to_them_take_the_chains_in_a_pass_of_their_own(handle, n, order):
    """Round 6: every other shape -- order 15 on those blocks, order 24 on other block sizes, ragged and very large blocks --
    is given the reference's chains too when no flag is set (two passes: orc_default_order_is_two_pass), everything equal to
    the oracle's ACORR_REFERENCE mode, floating point included; FLACENC_HIP_FLAG_CANONICAL_SUM_ORDER keeps the one-pass
    chunk tree there, which is NOT the reference's order on this corpus."""
    x = batch(4, n, 24, 99 + n + order)
    gp, gres, gR, gA = handle.qlpc_batch(x, 24, _capi.make_config(lpc_order=order), want_fp=True)
    cp, cres, cR, cA = orc.qlpc_batch(x, 24, orc.make_config(lpc_order=order, acorr=orc.ACORR_CANONICAL))
    rp, rres, rR, rA = orc.qlpc_batch(x, 24, orc.make_config(lpc_order=order, acorr=orc.ACORR_REFERENCE))
    assert np.array_equal(gR.view(np.uint64), rR.view(np.uint64)) and np.array_equal(gA.view(np.uint64), rA.view(np.uint64))
    assert np.array_equal(cR.view(np.uint64), rR.view(np.uint64))
    records_equal(gp, rp)
    assert np.array_equal(gres, rres)
    tp, tres, tR, tA = handle.qlpc_batch(x, 24, _capi.make_config(lpc_order=order, flags=_capi.FLAG_CANONICAL_SUM_ORDER), want_fp=True)
    op, ores, oR, oA = orc.qlpc_batch(x, 24, orc.make_config(lpc_order=order, acorr=orc.ACORR_CHUNK_TREE))
    assert np.array_equal(tR.view(np.uint64), oR.view(np.uint64))
    records_equal(tp, op)
    assert np.array_equal(tres, ores)
    assert not np.array_equal(oR.view(np.uint64), rR.view(np.uint64))


@pytest.mark.parametrize("kw", [dict(quant_precision=7), dict(window=("tukey", 1.0)), dict(window="rectangle"),
                                dict(rice_finest_only=True)])
def test_config_space(handle, kw):
    x = batch(4, 8192, 24, 31, namp=0.2)
    cfg = dict(kw)
    fin = cfg.pop("rice_finest_only", False)
    gcfg = _capi.make_config(lpc_order=24, rice_finest_only=fin, **cfg)
    gp, gres, _, _ = handle.qlpc_batch(x, 24, gcfg)
    op, ores, _, _ = orc.qlpc_batch(x, 24, orc.make_config(lpc_order=24, acorr=orc.ACORR_CANONICAL, rice_finest_only=fin, **cfg))
    records_equal(gp, op)
    assert np.array_equal(gres, ores)
