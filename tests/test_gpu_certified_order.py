"""The unflagged summation order on blocks of 4096 / 4608 samples at LPC orders up to 12 (the fused kernel's shapes) is
CERTIFIED: the kernels keep their own autocorrelation sums where the order certificate says the quantised parameters equal those of the
reference's sequential chains (src/lpc.rs:533-548) and recompute the subframe from those chains where they cannot prove it
(DESIGN.md section 2).  So with flags = 0:

  * every integer output -- QuantizedParameters, residual rows, Rice partitions, bit counts, decisions, frame bytes -- equals
    the oracle's ACORR_REFERENCE mode on 100 % of every corpus, the ill-conditioned ones included (near-pure sines, where
    the bare chunk tree is measurably different; DC + impulse; full-scale squares; the reference's real-audio fixtures);
  * everything, floating point included, equals the oracle's statement of the same rule (ACORR_CANONICAL on these shapes:
    orc_default_order_is_certified) bit for bit, and the device's counters (subframes analysed / certificates that needed
    the rows of T^-1 / subframes recomputed) equal the oracle's; launches of these shapes on other kernels (unaligned rows,
    FLACENC_HIP_FLAG_GENERIC_KERNEL) equal the oracle's ACORR_REFERENCE mode bit for bit;
  * FLACENC_HIP_FLAG_CANONICAL_SUM_ORDER gives the bare chunk tree (the oracle's ACORR_CHUNK_TREE).
"""
import numpy as np
import pytest

import util
from flacenc_rs_amd import _capi
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def handle():
    # (the hooks build -- the product library's objects + flacenc_hip_debug_*: every test here reads the certificate's
    # device counters or steers the order mode; the product library runs these shapes in test_gpu_parity.py)
    h = _capi.Handle(0, hooks=True)
    yield h
    h.close()


def gcfg(order, flags=0, **kw):
    return _capi.make_config(lpc_order=order, flags=flags, **kw)


def ocfg(order, acorr, **kw):
    return orc.make_config(lpc_order=order, acorr=acorr, **kw)


def records_equal(g, o, what=""):
    for f in ("order", "shift", "precision", "rice_order", "status", "code_bits", "subframe_bits", "sum_quotients"):
        assert np.array_equal(g[f], o[f]), (what, f, np.nonzero(g[f] != o[f])[0][:8])
    assert np.array_equal(g["coefs"], o["coefs"]), what
    assert np.array_equal(g["rice_params"], o["rice_params"]), what


# ---- corpora ---------------------------------------------------------------------------------------------------------
def near_pure_sines(count, n, bps=16, seed0=400):
    """Sine(57.3, 0.6) + Noise(5e-4): Toeplitz condition numbers of 1e7 and more -- the corpus on which the bare chunk tree's
    QuantizedParameters differ from the reference's in about one subframe of a few thousand."""
    return np.stack([util.sine_noise(n, bps, 57.3, 0.6, 5e-4, seed0 + i, phase=0.1 * i) for i in range(count)])


def noisy_sines(count, n, bps=16, seed0=100):
    return np.stack([util.sine_noise(n, bps, 36.0 + (i % 5), 0.4, 0.04, seed0 + i, phase=0.2 * i) for i in range(count)])


def dc_impulse(count, n, bps=16):
    out = np.zeros((count, n), np.int32)
    rng = np.random.default_rng(5)
    for i in range(count):
        out[i] = 1000 + 37 * i + rng.integers(-2, 3, n)
        out[i, (17 * i + 3) % n] = (1 << (bps - 1)) - 1
        if i % 3 == 0:
            out[i, i % 9] = -(1 << (bps - 1))  # an impulse in front of t = P: outside R[0]
    return out


def squares(count, n, bps=16):
    t = np.arange(n)
    hi, lo = (1 << (bps - 1)) - 1, -(1 << (bps - 1))
    return np.stack([np.where(((t + 3 * i) // (7 + i)) % 2 == 0, hi, lo).astype(np.int32) for i in range(count)])


def real_audio(n, step=256):
    rows = []
    for name in ("ras103", "ras22", "sus109", "sus6"):
        for ch in (0, 1):
            x = util.test_signal(name, ch)
            rows += [x[o:o + n] for o in range(0, 8192 - n + 1, step)]
    return np.stack(rows)


CORPORA = {
    "near_pure_sines": lambda n: near_pure_sines(96, n),
    "noisy_sines": lambda n: noisy_sines(64, n),
    "dc_impulse": lambda n: dc_impulse(24, n),
    "squares": lambda n: squares(16, n),
    "real_audio": lambda n: real_audio(n),
    "silence_and_constants": lambda n: np.stack([np.zeros(n, np.int32), np.full(n, -7, np.int32), np.full(n, 32767, np.int32),
                                                 np.arange(n, dtype=np.int32) % 5 - 2]),
}


def certified_exact(handle, x, bps, order, flags=0, **kw):
    """GPU (flags = 0) == oracle's certified rule bit for bit, == oracle's reference order on every integer output."""
    rule = orc.ACORR_CANONICAL
    import torch
    x = np.ascontiguousarray(x, np.int32)
    stats = torch.zeros(3, dtype=torch.int32, device="cuda")
    handle.debug_set_cert_stats(stats.data_ptr())
    try:
        gp, gres, gR, gA = handle.qlpc_batch(x, bps, gcfg(order, flags=flags, **kw), want_fp=True)
        torch.cuda.synchronize()
    finally:
        handle.debug_set_cert_stats(0)
    orc.cert_stats(reset=True)
    cp, cres, cR, cA = orc.qlpc_batch(x, bps, ocfg(order, rule, **kw), nthreads=1)
    want_stats = orc.cert_stats()
    rp, rres, rR, rA = orc.qlpc_batch(x, bps, ocfg(order, orc.ACORR_REFERENCE, **kw))
    # the rule, floating point included
    assert np.array_equal(gR.view(np.uint64), cR.view(np.uint64)), "R[] against the oracle's certified rule"
    assert np.array_equal(gA.view(np.uint64), cA.view(np.uint64)), "LPC coefficient bits against the oracle's certified rule"
    records_equal(gp, cp, "certified rule")
    assert np.array_equal(gres, cres)
    # what the rule is for: the reference's integers, all of them
    same = (gp["coefs"] == rp["coefs"]).all(axis=1) & (gp["shift"] == rp["shift"]) & (gp["order"] == rp["order"])
    assert int(same.sum()) == same.size, (int(same.sum()), same.size)
    records_equal(gp, rp, "reference order")
    assert np.array_equal(gres, rres)
    got = tuple(int(v) for v in stats.cpu().numpy())
    return got, want_stats


@pytest.mark.parametrize("order", [8, 10, 12])
@pytest.mark.parametrize("corpus", sorted(CORPORA))
def test_default_order_is_the_references_on_4096(handle, corpus, order):
    x = CORPORA[corpus](4096)
    got, want = certified_exact(handle, x, 16, order)
    print(f"{corpus}, order {order}: {want[0]} subframes, {want[1]} certificates needed the rows of T^-1, {want[2]} recomputed "
          f"from the reference's chains; reference-identical fraction 1.0")
    assert got == want, (got, want)


@pytest.mark.parametrize("order", [1, 2, 5, 9, 11])
def test_other_orders_and_the_4608_block(handle, order):
    got, want = certified_exact(handle, near_pure_sines(24, 4096, seed0=900), 16, order)
    assert got == want
    x = np.concatenate([near_pure_sines(12, 4608, seed0=77), noisy_sines(12, 4608)])
    got, want = certified_exact(handle, x, 16, order)
    assert got == want


def test_systems_that_are_not_positive_definite_are_recomputed(handle):
    """Round 6 (tests/test_certificate_cpu.py::test_a_system_that_is_not_positive_definite_is_not_certified): lag sums from
    t = P on need not be an autocorrelation; where the recursion meets a non-positive denominator it is unstable, the
    certificate does not apply and the subframe takes the reference's chains -- on the device as in the oracle, counters
    included.  The attack's 24-bit counter-example, and plateaus in front of 16-bit material (same mechanism)."""
    import os
    x = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cert_nonpd_plateau_24bit.npy"))
    got, want = certified_exact(handle, np.stack([x, x, x, x]), 24, 8, window="rectangle")
    assert got == want == (4, 0, 4), (got, want)
    y = noisy_sines(16, 4096, seed0=31)
    for k in range(16):
        y[k, : 8 + k] = 30000 if k % 2 else -30000  # a full-scale plateau over the first lags
    for window in ("rectangle", ("tukey", 0.4)):
        for order in (8, 12):
            got, want = certified_exact(handle, y, 16, order, window=window)
            assert got == want, (window, order, got, want)


@pytest.mark.parametrize("n", [256, 576, 1024, 1152, 2048, 2304])
@pytest.mark.parametrize("order", [8, 12])
def test_sub_wave_shapes_take_the_references_chains(handle, n, order):
    """Round 6: on blocks of 256 .. 2304 samples (the sub-wave kernel) the unflagged order IS the reference's: its chains
    for every subframe on the matrix cores in front (acorr_reference_mfma_kernel), the kernel's own autocorrelation skipped.
    Everything -- R[], unquantised and quantised coefficients, records, rows -- equals the oracle's ACORR_REFERENCE mode (and
    its ACORR_CANONICAL, which states the same rule) on easy material, on hard material and on a mixture; no certificate
    runs (the counters stay at zero), and FLACENC_HIP_FLAG_CANONICAL_SUM_ORDER keeps the one-pass chunk tree.  (An order
    certificate inside the kernel was built first and dropped: 20 to 140 x slower on music at orders 10-12, and its int16-image
    instance mis-stated max |s| for blocks holding -32768 -- found by tools/fuzz_subwave.py.)"""
    import torch
    easy = noisy_sines(40, n, seed0=10 + order)
    hard = near_pure_sines(40, n, seed0=500 + order)
    mixed = np.concatenate([easy[:13], hard[:14], easy[13:20], dc_impulse(6, n), np.zeros((2, n), np.int32)])
    full = np.full((3, n), -32768, np.int32)
    full[1, ::2] = 32767
    full[2, n // 3:] = 0
    for x, bps, window in ((easy, 16, ("tukey", 0.4)), (hard, 16, ("tukey", 0.4)), (mixed, 16, ("tukey", 0.4)),
                           (mixed, 24, "rectangle"), (np.concatenate([full, hard[:5]]), 16, ("tukey", 0.4))):
        stats = torch.zeros(3, dtype=torch.int32, device="cuda")
        handle.debug_set_cert_stats(stats.data_ptr())
        try:
            gp, gres, gR, gA = handle.qlpc_batch(x, bps, gcfg(order, window=window), want_fp=True)
            torch.cuda.synchronize()
        finally:
            handle.debug_set_cert_stats(0)
        assert stats.cpu().tolist() == [0, 0, 0]
        for mode in (orc.ACORR_REFERENCE, orc.ACORR_CANONICAL):
            rp, rres, rR, rA = orc.qlpc_batch(x, bps, ocfg(order, mode, window=window))
            assert np.array_equal(gR.view(np.uint64), rR.view(np.uint64)) and np.array_equal(gA.view(np.uint64), rA.view(np.uint64))
            records_equal(gp, rp, "sub-wave shape, unflagged")
            assert np.array_equal(gres, rres)
    tp, tres, tR, tA = handle.qlpc_batch(hard, 16, gcfg(order, flags=_capi.FLAG_CANONICAL_SUM_ORDER), want_fp=True)
    op, ores, oR, oA = orc.qlpc_batch(hard, 16, ocfg(order, orc.ACORR_CHUNK_TREE))
    assert np.array_equal(tR.view(np.uint64), oR.view(np.uint64)) and np.array_equal(tres, ores)
    records_equal(tp, op, "sub-wave shape, chunk tree")
    rp, _, rR, _ = orc.qlpc_batch(hard, 16, ocfg(order, orc.ACORR_REFERENCE))
    assert not np.array_equal(oR[:, :order + 1].view(np.uint64), rR[:, :order + 1].view(np.uint64))  # (the orders do differ here)


@pytest.mark.parametrize("n,use_fixed", [(1152, True), (2304, False), (512, True)])
def test_sub_wave_frames_are_the_references(handle, n, use_fixed):
    """... and through encode_frame on these shapes (the reference's chains for the four roles' QLPC candidates in front of
    qlpc_subwave_kernel's frame variant): decision records, chosen rows and packed bytes of every frame are the oracle's in
    the REFERENCE order, hard frames and easy ones side by side -- stereo, and 8 independent channels through the
    int16-image instance."""
    nf = 96
    rng = np.random.default_rng(n)
    base = np.concatenate([near_pure_sines(16, n, seed0=40), noisy_sines(16, n, seed0=60)])
    pick = rng.integers(0, 32, size=(nf, 2))
    frames = np.ascontiguousarray(np.stack([base[pick[:, 0]], base[pick[:, 1]]], axis=1))
    fc = _capi.make_frame_config(gcfg(8), use_fixed=use_fixed)
    res, resid = handle.encode_stereo_frames(frames, 16, fc)
    blobs = handle.pack_stereo_frames(frames, res, resid, 16, 44100)
    ofc = orc.make_frame_config(ocfg(8, orc.ACORR_REFERENCE), use_fixed=use_fixed)
    ores, oresid = orc.encode_stereo_frames_cfg(frames, 16, ofc)
    assert res["channel_assignment"].tolist() == ores["channel_assignment"].tolist()
    assert res["kind"].tolist() == ores["kind"].tolist() and res["bits"].tolist() == ores["bits"].tolist()
    assert np.array_equal(resid, oresid)
    for f in range(nf):
        assert res[f] == ores[f], f
        assert blobs[f] == orc.write_stereo_frame(ores[f], frames[f, 0], frames[f, 1], 16, 44100, f, oresid[f, 0], oresid[f, 1]), f
    # 8 independent channels (flacenc_hip_encode_frames: the sub-wave kernel's independent-channel variant)
    ch = np.ascontiguousarray(base[rng.integers(0, 32, size=(24, 8))])
    cres, cresid = handle.encode_frames(ch, 16, fc)
    want = [orc.encode_subframe(ch[f, c], 16, ofc) for f in range(24) for c in range(8)]
    for i, w in enumerate(want):
        f, c = divmod(i, 8)
        assert int(cres[f, c]["kind"]) == w["kind"] and int(cres[f, c]["bits"]) == w["bits"], (f, c)
        if w["kind"] >= 2:
            assert np.array_equal(cresid[f, c], w["residual"]), (f, c)


def test_the_corpus_separates_the_orders(handle):
    """The bare chunk tree (FLACENC_HIP_FLAG_CANONICAL_SUM_ORDER) is NOT the reference's on this corpus -- R[] never, the
    quantised parameters in a few subframes of some thousands -- so the equalities above are not vacuous."""
    x = near_pure_sines(1200, 4096, seed0=400)
    tp, tres, tR, tA = handle.qlpc_batch(x, 16, gcfg(8, flags=_capi.FLAG_CANONICAL_SUM_ORDER), want_fp=True)
    op, ores, oR, oA = orc.qlpc_batch(x, 16, ocfg(8, orc.ACORR_CHUNK_TREE))
    assert np.array_equal(tR.view(np.uint64), oR.view(np.uint64))
    records_equal(tp, op, "chunk tree")
    assert np.array_equal(tres, ores)
    rp, rres, rR, rA = orc.qlpc_batch(x, 16, ocfg(8, orc.ACORR_REFERENCE))
    assert not np.array_equal(tR[:, :9].view(np.uint64), rR[:, :9].view(np.uint64))
    differ = int(((tp["coefs"] != rp["coefs"]).any(axis=1) | (tp["shift"] != rp["shift"])).sum())
    print(f"bare chunk tree: QuantizedParameters differ from the reference's in {differ} of {x.shape[0]} near-pure-sine subframes")
    got, want = certified_exact(handle, x, 16, 8)
    assert got == want and want[2] > 0


def test_other_kernels_on_these_shapes_take_the_references_order(handle):
    """The default on these shapes is "the reference's integers": launches the fused kernel cannot take --
    FLACENC_HIP_FLAG_GENERIC_KERNEL, rows whose stride is not a multiple of four samples -- are given the reference's own
    chains (acorr_reference_kernel) outright: everything equals the oracle's ACORR_REFERENCE mode, floating point included."""
    import torch
    x = np.concatenate([near_pure_sines(10, 4096, seed0=31), noisy_sines(6, 4096), real_audio(4096, step=2048)])
    gp, gres, gR, gA = handle.qlpc_batch(x, 16, gcfg(10, flags=_capi.FLAG_GENERIC_KERNEL), want_fp=True)
    rp, rres, rR, rA = orc.qlpc_batch(x, 16, ocfg(10, orc.ACORR_REFERENCE))
    assert np.array_equal(gR.view(np.uint64), rR.view(np.uint64)) and np.array_equal(gA.view(np.uint64), rA.view(np.uint64))
    records_equal(gp, rp, "generic kernel")
    assert np.array_equal(gres, rres)
    ns, n = x.shape
    stride = n + 3
    buf = torch.zeros(ns * stride + 1, dtype=torch.int32, device="cuda")
    rows = buf[1:].view(ns, stride)  # base misaligned by one sample
    rows[:, :n] = torch.from_numpy(x).cuda()
    params = torch.zeros((ns, 352), dtype=torch.uint8, device="cuda")
    resid = torch.zeros((ns, n), dtype=torch.int32, device="cuda")
    bps = torch.full((ns,), 16, dtype=torch.uint8, device="cuda")
    handle.qlpc_batch_device(gcfg(10), rows.data_ptr(), ns, n, stride, bps.data_ptr(), params.data_ptr(), resid.data_ptr(), n,
                             sync=True)
    torch.cuda.synchronize()
    gp = np.frombuffer(params.cpu().numpy().tobytes(), dtype=_capi.PARAMS_DTYPE)
    records_equal(gp, rp, "unaligned rows, reference order")
    assert np.array_equal(resid.cpu().numpy(), rres)


@pytest.mark.parametrize("n", [4096, 4608])
@pytest.mark.parametrize("use_fixed", [False, True])
def test_frames_and_bytes_are_the_references(handle, n, use_fixed):
    """encode_frame's decisions, the chosen rows and the packed frame bytes with flags = 0 == the oracle in the reference's
    order (stereo roles: the four candidates of a frame are certified one by one)."""
    l = np.concatenate([near_pure_sines(10, n, seed0=5), noisy_sines(6, n, seed0=8)])
    r = np.concatenate([near_pure_sines(10, n, seed0=50), noisy_sines(6, n, seed0=80)])
    frames = np.stack([l, r], axis=1)
    fc = _capi.make_frame_config(gcfg(10), use_fixed=use_fixed)
    res, resid = handle.encode_stereo_frames(frames, 16, fc)
    blobs = handle.pack_stereo_frames(frames, res, resid, 16, 44100)
    ofc = orc.make_frame_config(ocfg(10, orc.ACORR_REFERENCE), use_fixed=use_fixed)
    ores, oresid = orc.encode_stereo_frames_cfg(frames, 16, ofc)
    for f in range(frames.shape[0]):
        want = orc.write_stereo_frame(ores[f], frames[f, 0], frames[f, 1], 16, 44100, f, oresid[f, 0], oresid[f, 1])
        assert blobs[f] == want, f
    assert np.array_equal(resid, oresid)


def test_fused_bit_writer_takes_the_references_chains(handle):
    """FLACENC_HIP_FLAG_FUSED_PACK (bytes only, no certificate inside that kernel): launch_qlpc hands it the reference's
    R[], so its bytes are the two-kernel form's and the oracle's."""
    import torch
    n, nf = 4096, 12
    l = near_pure_sines(nf, n, seed0=15)
    r = near_pure_sines(nf, n, seed0=51)
    frames = np.ascontiguousarray(np.stack([l, r], axis=1))
    x = torch.from_numpy(frames).cuda()
    out_stride = handle.frame_bytes_bound(n, 16)
    ofc = orc.make_frame_config(ocfg(8, orc.ACORR_REFERENCE), use_fixed=True)
    ores, oresid = orc.encode_stereo_frames_cfg(frames, 16, ofc)
    for flag in (_capi.FLAG_FUSED_PACK, _capi.FLAG_TWO_STAGE_PACK):
        fc = _capi.make_frame_config(gcfg(8, flags=flag), use_fixed=True)
        results = torch.zeros((nf, 752), dtype=torch.uint8, device="cuda")
        packed = torch.zeros((nf, out_stride), dtype=torch.uint8, device="cuda")
        lens = torch.zeros(nf, dtype=torch.int32, device="cuda")
        handle.encode_pack_stereo_frames_device(fc, x.data_ptr(), nf, n, n, 16, 44100, 0, 1, results.data_ptr(),
                                                packed.data_ptr(), out_stride, lens.data_ptr())
        torch.cuda.synchronize()
        pk, ln = packed.cpu().numpy(), lens.cpu().numpy()
        for f in range(nf):
            want = orc.write_stereo_frame(ores[f], frames[f, 0], frames[f, 1], 16, 44100, f, oresid[f, 0], oresid[f, 1])
            assert bytes(pk[f, :ln[f]]) == want, (flag, f)


def test_independent_channels_are_certified_too(handle):
    """Plain batches and Independent(n) frames (four subframes of one workgroup certified side by side), 4096 and 4608."""
    for n in (4096, 4608):
        chans = np.concatenate([near_pure_sines(5, n, seed0=61), noisy_sines(3, n, seed0=62)])
        frames = chans.reshape(2, 4, n)
        fc = _capi.make_frame_config(gcfg(10), use_fixed=True)
        res, resid = handle.encode_frames(frames, 16, fc)
        ofc = orc.make_frame_config(ocfg(10, orc.ACORR_REFERENCE), use_fixed=True)
        for f in range(frames.shape[0]):
            for c in range(frames.shape[1]):
                w = orc.encode_subframe(frames[f, c], 16, ofc)
                g = res[f, c]
                assert int(g["kind"]) == w["kind"] and int(g["bits"]) == w["bits"], (n, f, c)
                if w["kind"] >= 2:
                    assert np.array_equal(resid[f, c], w["residual"]), (n, f, c)


def test_integer_parity_only_keeps_the_certified_order_and_the_references_integers(handle):
    """FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER | _INTEGER_PARITY_ONLY (what the Rust / C++ drop-in passes for a stable build):
    on the certified shapes no second pass -- the certificate's counters move, R[] is the kernel's own where certified --
    and the integers are the reference's; with the fixed-LPC candidate and on a shape that is not certified (a ragged
    block) everything is the plain reference-order mode's."""
    import torch
    flags = _capi.FLAG_REFERENCE_SUM_ORDER | _capi.FLAG_INTEGER_PARITY_ONLY
    x = np.concatenate([near_pure_sines(12, 4096, seed0=71), noisy_sines(20, 4096, seed0=72)])
    stats = torch.zeros(3, dtype=torch.int32, device="cuda")
    handle.debug_set_cert_stats(stats.data_ptr())
    try:
        gp, gres, gR, gA = handle.qlpc_batch(x, 16, gcfg(8, flags=flags), want_fp=True)
        torch.cuda.synchronize()
    finally:
        handle.debug_set_cert_stats(0)
    assert int(stats[0]) == x.shape[0]  # certified inside the fused kernel, not by a pass in front of it
    cp, cres, cR, cA = orc.qlpc_batch(x, 16, ocfg(8, orc.ACORR_CANONICAL))
    rp, rres, rR, rA = orc.qlpc_batch(x, 16, ocfg(8, orc.ACORR_REFERENCE))
    assert np.array_equal(gR.view(np.uint64), cR.view(np.uint64))  # (the unflagged rule's floating point)
    records_equal(gp, rp, "integers")
    assert np.array_equal(gres, rres)
    # frames with the default candidates: bytes == the oracle in the reference's orders (selector sums included)
    l, r = near_pure_sines(6, 4096, seed0=81), noisy_sines(6, 4096, seed0=82)
    frames = np.stack([l, r], axis=1)
    fc = _capi.make_frame_config(gcfg(10, flags=flags), use_fixed=True)
    res, resid = handle.encode_stereo_frames(frames, 16, fc)
    blobs = handle.pack_stereo_frames(frames, res, resid, 16, 44100)
    ofc = orc.make_frame_config(ocfg(10, orc.ACORR_REFERENCE), use_fixed=True, fixed=orc.make_fixed_config(sum_mode=orc.SUMABS_STABLE))
    ores, oresid = orc.encode_stereo_frames_cfg(frames, 16, ofc)
    for f in range(frames.shape[0]):
        assert blobs[f] == orc.write_stereo_frame(ores[f], frames[f, 0], frames[f, 1], 16, 44100, f, oresid[f, 0], oresid[f, 1]), f
    # a shape without a certificate: the flag pair is the plain reference-order mode
    y = noisy_sines(6, 1000, seed0=90)
    p1, r1, R1, A1 = handle.qlpc_batch(y, 16, gcfg(8, flags=flags), want_fp=True)
    p2, r2, R2, A2 = handle.qlpc_batch(y, 16, gcfg(8, flags=_capi.FLAG_REFERENCE_SUM_ORDER), want_fp=True)
    assert np.array_equal(R1.view(np.uint64), R2.view(np.uint64)) and np.array_equal(r1, r2)
    records_equal(p1, p2, "ragged block")


@pytest.mark.parametrize("order,use_fixed,nf", [(8, False, 1024), (10, True, 1024), (12, False, 160)])
def test_two_pass_form_on_hard_material_gives_the_same_bytes(handle, order, use_fixed, nf):
    """Integer-only launches watch the certificate's counters (a verdict per 4096 subframes) and take the two-pass form (the
    reference's chains for every subframe on the matrix cores, then the fused kernel on their R[]) while the material last
    seen was hard (flacenc_hip_api.cpp, launch_adaptive).  A choice of speed, never of result: on a batch of near-pure
    tones every launch -- certified kernel, two-pass, probe -- writes the same records and rows, the oracle's in the
    reference's order; a noisy batch never leaves the certified kernel.  Launches of 640 subframes: their counters add up to
    a verdict, the probe takes two of them."""
    import torch
    n = 4096
    rng = np.random.default_rng(order)
    base_l, base_r = near_pure_sines(24, n, seed0=700 + order), near_pure_sines(24, n, seed0=800 + order)
    pick = rng.integers(0, 24, size=(nf, 2))
    hard = np.ascontiguousarray(np.stack([base_l[pick[:, 0]], base_r[pick[:, 1]]], axis=1))
    easy_l, easy_r = noisy_sines(24, n, seed0=900), noisy_sines(24, n, seed0=950)
    easy = np.ascontiguousarray(np.stack([easy_l[pick[:, 0]], easy_r[pick[:, 1]]], axis=1))
    fc = _capi.make_frame_config(gcfg(order), use_fixed=use_fixed)
    ofc = orc.make_frame_config(ocfg(order, orc.ACORR_REFERENCE), use_fixed=use_fixed)
    results = torch.zeros((nf, _capi.FRAME_RESULT_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
    residual = torch.zeros((nf * 2, n), dtype=torch.int32, device="cuda")

    def run(x):
        results.zero_()
        residual.zero_()
        handle.encode_stereo_frames_device(fc, x.data_ptr(), nf, n, n, 16, results.data_ptr(), residual.data_ptr(), n)
        torch.cuda.synchronize()
        return results.cpu().numpy().tobytes(), residual.cpu().numpy().copy()

    handle.debug_set_adaptive_order(True)  # (also resets the state a previous test left)
    try:
        xe = torch.from_numpy(easy).cuda()
        ref = run(xe)
        for _ in range(4):
            got = run(xe)
            assert got[0] == ref[0] and np.array_equal(got[1], ref[1])
            assert handle.debug_adaptive_state() == (0, 0), "a noisy batch stays on the certified kernel"
        xh = torch.from_numpy(hard).cuda()
        first = run(xh)  # certified kernel (nothing known about this material yet)
        spans = []
        for _ in range(30):  # span of 8, probe, span of 16, probe ...
            got = run(xh)
            assert got[0] == first[0] and np.array_equal(got[1], first[1])
            spans.append(handle.debug_adaptive_state()[0])
        assert 8 in spans and 16 in spans, spans
        if nf * 4 >= 4096:
            assert spans[0] == 8, spans  # (one launch is a verdict; smaller launches add up to one)
        handle.debug_set_adaptive_order(False)
        pinned = run(xh)
        assert pinned[0] == first[0] and np.array_equal(pinned[1], first[1])
    finally:
        handle.debug_set_adaptive_order(True)
    # ... and they are the oracle's in the reference's order (the distinct frames: every pair of the 24 + 24 base channels used)
    g = np.frombuffer(first[0], dtype=_capi.FRAME_RESULT_DTYPE)
    rows = first[1].reshape(nf, 2, n)
    seen = {}
    for f in range(nf):
        seen.setdefault((int(pick[f, 0]), int(pick[f, 1])), f)
    idx = sorted(seen.values())[:48]
    want, wrows = orc.encode_stereo_frames_cfg(hard[idx], 16, ofc)
    for k, f in enumerate(idx):
        assert g[f] == want[k], f
        assert np.array_equal(rows[f], wrows[k]), f
