"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle.

Parity contract (DESIGN.md "Parity"):
  T0  GPU == oracle(canonical summation order) bit-for-bit on EVERYTHING, including the f64
      autocorrelation and LPC coefficients.
  T2  GPU vs oracle(reference summation order, src/lpc.rs:533-548), stated fp tolerance:
      |dR| <= 1e-12 * R[0] on the autocorrelation; |da| <= 1e-5 * |a| + 1e-5 on the LPC
      coefficients (the reference's own `assert_close!`, src/test_helper.rs:46-56, which is
      all its simd-vs-nosimd parity test asserts, src/lpc.rs:1392-1413 -- Levinson amplifies
      the 1e-16-level reordering noise by the Toeplitz condition number, up to ~1e-7 relative
      on order-24 real audio).  Wherever the quantised coefficients agree, residual / Rice
      partitions / bit counts are bit-equal.
  T3  every emitted subframe decodes (decode.rs:159-177) to its input.
  Integer stages alone: oracle integer stages fed with the GPU's own quantised coefficients
  reproduce the GPU residual and Rice search on ALL subframes.
"""
import os
import zlib

import numpy as np
import pytest

import util
from flacenc_rs_amd import _capi
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

ACORR_RTOL = 1e-12              # relative to R[0]
LPC_RTOL, LPC_ATOL = 1e-5, 1e-5  # assert_close!, src/test_helper.rs:46-56


@pytest.fixture(scope="module")
def handle():
    h = _capi.Handle(0)
    yield h
    h.close()


def gpu_cfg(order, precision=15, window=("tukey", 0.4), max_p=30):
    return _capi.make_config(lpc_order=order, quant_precision=precision, window=window,
                             max_rice_parameter=max_p)


def orc_cfg(order, precision=15, window=("tukey", 0.4), max_p=30, acorr=orc.ACORR_REFERENCE):
    return orc.make_config(lpc_order=order, quant_precision=precision, window=window,
                           max_rice_parameter=max_p, acorr=acorr)


def assert_records_equal(g, o, what=""):
    for f in ("order", "shift", "precision", "rice_order", "status", "code_bits", "subframe_bits",
              "sum_quotients"):
        assert np.array_equal(g[f], o[f]), (what, f, g[f][:8], o[f][:8])
    assert np.array_equal(g["coefs"], o["coefs"]), what
    assert np.array_equal(g["rice_params"], o["rice_params"]), what


def check_integer_stages_from_gpu_coefs(x, bps, params, residual, max_p):
    """oracle compute_error + PRC search + bit counts on the GPU's quantised parameters."""
    for k in range(x.shape[0]):
        p = params[k]
        order = int(p["order"])
        qp = orc.qparams(p["coefs"][:order], int(p["shift"]), int(p["precision"]))
        e = orc.compute_error(qp, x[k])
        assert np.array_equal(e, residual[k]), k
        res = orc.encode_residual(e, order, max_p)
        assert res["partition_order"] == int(p["rice_order"]), k
        np_ = 1 << res["partition_order"]
        assert res["rice_params"].tolist() == p["rice_params"][:np_].tolist(), k
        assert (p["rice_params"][np_:] == 0).all()
        assert res["code_bits"] == int(p["code_bits"]), k
        assert res["sum_quotients"] == int(p["sum_quotients"]), k
        bits = orc.lpc_count_bits(int(np.broadcast_to(bps, (x.shape[0],))[k]), order,
                                  int(p["precision"]), res["count_bits"])
        assert bits == int(p["subframe_bits"]), k


def check_lossless(x, params, residual):
    for k in range(x.shape[0]):
        p = params[k]
        order = int(p["order"])
        dec = orc.decode_lpc(x[k][:order], p["coefs"][:order], int(p["shift"]), residual[k])
        assert np.array_equal(dec, x[k]), k
        assert (residual[k][:order] == 0).all()


def full_parity(handle, x, bps, order, precision=15, window=("tukey", 0.4), max_p=30,
                integer_check=True, coef_tolerance=True):
    x = np.ascontiguousarray(x, np.int32)
    gp, gres, gR, gA = handle.qlpc_batch(x, bps, gpu_cfg(order, precision, window, max_p), want_fp=True)
    assert (gp["status"] == 0).all()
    # T0: canonical order, everything bit-exact
    cp, cres, cR, cA = orc.qlpc_batch(x, bps, orc_cfg(order, precision, window, max_p, orc.ACORR_CANONICAL))
    assert np.array_equal(gR.view(np.uint64), cR.view(np.uint64)), "autocorrelation bits (canonical)"
    assert np.array_equal(gA.view(np.uint64), cA.view(np.uint64)), "LPC coefficient bits (canonical)"
    assert_records_equal(gp, cp, "canonical")
    assert np.array_equal(gres, cres)
    # T2: reference order within the stated tolerance; integers equal where coefficients agree
    rp, rres, rR, rA = orc.qlpc_batch(x, bps, orc_cfg(order, precision, window, max_p))
    r0 = np.maximum(np.abs(rR[:, :1]), 1e-300)
    assert (np.abs(gR - rR) / r0 <= ACORR_RTOL).all()
    same = (gp["coefs"] == rp["coefs"]).all(axis=1) & (gp["shift"] == rp["shift"])
    n_sub, n_same = int(same.size), int(same.sum())
    certified_shape = orc.default_order_is_certified(x.shape[1], order)
    stable_shape = x.shape[1] in (4096, 8192, 16384) and order >= 16
    # the measured fraction of subframes whose QuantizedParameters are the reference's, per corpus (pytest -s / -rP)
    print(f"reference-identical QuantizedParameters: {n_same} of {n_sub} subframes "
          f"(block {x.shape[1]}, order {order}, {'certified' if certified_shape else 'stable order' if stable_shape else 'chunk tree'})")
    if certified_shape or stable_shape:
        # the unflagged order on these shapes is certified against (or IS) the reference's: all of them, exactly
        assert n_same == n_sub, (n_same, n_sub)
        assert_records_equal(gp, rp, "reference order")
        assert np.array_equal(gres, rres)
    if coef_tolerance:
        assert (np.abs(gA - rA) <= LPC_RTOL * np.abs(rA) + LPC_ATOL).all()
        assert n_same >= 0.99 * n_sub, (n_same, n_sub)
    assert_records_equal(gp[same], rp[same], "reference order, same coefficients")
    assert np.array_equal(gres[same], rres[same])
    # integer stages alone + T3
    if integer_check:
        check_integer_stages_from_gpu_coefs(x, bps, gp, gres, max_p)
    check_lossless(x, gp, gres)
    return gp, gres


def batch_sine_noise(ns, n, bps, seed0=1000):
    return np.stack([util.sine_noise(n, bps, 20 + 13 * (k % 17), 0.1 + 0.05 * (k % 9),
                                     0.01 * (1 + k % 11), seed=seed0 + k, phase=0.1 * k)
                     for k in range(ns)])


# ------------------------------------------------------------------ BASELINE configs ----
def test_config2_44k_16bit_order8(handle):
    """BASELINE config 2: 16-bit, block 4096, LPC order 8 (full Rice search)."""
    x = batch_sine_noise(48, 4096, 16)
    full_parity(handle, x, 16, 8)


def test_config1_default_order10_with_side_channel(handle):
    """BASELINE config 1 shape: default config::Encoder (order 10, precision 15, Tukey 0.4),
    stereo L, R, M, S analysed as four subframes; side carries bps + 1 (coding.rs:444)."""
    l = batch_sine_noise(12, 4096, 16, seed0=1)
    r = batch_sine_noise(12, 4096, 16, seed0=500)
    ms = [orc.stereo_to_midside(a, b) for a, b in zip(l, r)]
    m = np.stack([v[0] for v in ms])
    s = np.stack([v[1] for v in ms])
    x = np.concatenate([l, r, m, s])
    bps = np.array([16] * 36 + [17] * 12, np.uint8)
    full_parity(handle, x, bps, 10)


def test_config3_96k_24bit_order24_and_32(handle):
    """BASELINE config 3: 24-bit, block 8192, order 24 (reference max) and 32 (extension)."""
    x = batch_sine_noise(6, 8192, 24, seed0=7)
    full_parity(handle, x, 24, 24)
    full_parity(handle, x, 24, 32)


def test_config4_8channel_order10(handle):
    """BASELINE config 4: 8 independent channels of 16-bit, block 4096."""
    x = batch_sine_noise(16, 4096, 16, seed0=90)
    full_parity(handle, x, 16, 10)


def test_config5_block16384_24bit_order24_and_32(handle):
    """BASELINE config 5: 24-bit, block 16384, orders 24 / 32."""
    x = batch_sine_noise(4, 16384, 24, seed0=55)
    full_parity(handle, x, 24, 24)
    full_parity(handle, x[:2], 25, 32)


# ------------------------------------------------------------------ golden fixtures ----
GOLD = np.load(os.path.join(util.GOLDEN, "qlpc_golden.npz"))
GOLD_NAMES = sorted({k.split("/")[0] for k in GOLD.files})


@pytest.mark.parametrize("name", GOLD_NAMES)
def test_golden_vectors(handle, name):
    """Committed expectations (oracle, reference order): fp within tolerance, integers equal."""
    n, order, bps = (int(v) for v in GOLD[f"{name}/meta"])
    x = GOLD[f"{name}/input"].astype(np.int32)[None, :]
    gp, gres, gR, gA = handle.qlpc_batch(x, bps, gpu_cfg(order), want_fp=True)
    R = GOLD[f"{name}/autocorr"]
    A = GOLD[f"{name}/lpc_coefs"]
    assert (np.abs(gR[0, : order + 1] - R) / abs(R[0]) <= ACORR_RTOL).all()
    assert (np.abs(gA[0, :order] - A) <= LPC_RTOL * np.abs(A) + LPC_ATOL).all()
    p = gp[0]
    sc = GOLD[f"{name}/scalars"].tolist()
    assert p["coefs"][: sc[0]].tolist() == GOLD[f"{name}/coefs"].tolist()
    assert [int(p["order"]), int(p["shift"]), int(p["rice_order"]), int(p["code_bits"]),
            int(p["subframe_bits"]), int(p["sum_quotients"]), zlib.crc32(gres[0].tobytes())] == sc
    assert p["rice_params"][: 1 << sc[2]].tolist() == GOLD[f"{name}/rice_params"].tolist()


def test_reference_fixture_signals(handle):
    """The reference's real-audio fixtures (src/resource/*.bin), all eight, as 4096-blocks."""
    blocks = []
    for name in ("sus109", "sus6", "ras22", "ras103"):
        for ch in (0, 1):
            sig = util.test_signal(name, ch)
            blocks += [sig[:4096], sig[4096:]]
    full_parity(handle, np.stack(blocks), 16, 10)
    full_parity(handle, np.stack(blocks), 16, 8, precision=12, window=("tukey", 0.1))  # lpc.rs:1258-1295


# ------------------------------------------------------------------ edge cases ----
@pytest.mark.parametrize("n", [64, 65, 100, 192, 576, 1000, 1001, 1152, 2304, 4608, 4095, 4097, 8191])
def test_ragged_block_sizes(handle, n):
    """Tail blocks of any length >= 64 reach the path (coding.rs:396); the finest
    partition order then depends on trailing zeros of n (rice.rs:157-165)."""
    x = batch_sine_noise(5, n, 16, seed0=n)
    full_parity(handle, x, 16, 10)


@pytest.mark.parametrize("n", [1040, 1152, 1280, 1281, 2064, 2304, 2400, 2560, 2561, 4095])
@pytest.mark.parametrize("order", [4, 8, 12, 16, 32])
def test_blocks_a_little_over_half_a_workgroup(handle, n, order):
    """plan_qlpc_launch: a block whose 16-sample rows fill between 1/2 and 5/8 of the next power-of-two workgroup
    (1152 = 72 rows, 2304 = 144) runs on half of it in two chunk rounds, the second mostly idle -- in every order
    bucket (their workgroup limits differ), either side of the 5/8 boundary (1280 | 1281, 2560 | 2561), 24-bit too."""
    full_parity(handle, batch_sine_noise(3, n, 16, seed0=n + order), 16, order)
    full_parity(handle, batch_sine_noise(2, n, 24, seed0=n - order), 24, order)


@pytest.mark.parametrize("n", [16384 + 256, 24576, 32512, 32767])
def test_maximum_block_sizes(handle, n):
    """Up to MAX_BLOCK_SIZE = 32767 (constant.rs:57): the unpadded-LDS kernel variant."""
    x = batch_sine_noise(2, n, 16, seed0=n)
    full_parity(handle, x, 16, 8, integer_check=(n != 24576))
    full_parity(handle, x[:1], 16, 24, integer_check=False)


@pytest.mark.parametrize("order", [1, 2, 3, 5, 7, 9, 11, 12, 13, 16, 17, 20, 24, 25, 31, 32])
def test_every_order_bucket(handle, order):
    x = batch_sine_noise(4, 4096, 16, seed0=order)
    full_parity(handle, x, 16, order)


@pytest.mark.parametrize("precision", [1, 2, 5, 12, 15])
def test_quantisation_precisions(handle, precision):
    x = batch_sine_noise(4, 4096, 16, seed0=precision)
    full_parity(handle, x, 16, 10, precision=precision)


@pytest.mark.parametrize("window", ["rectangle", ("tukey", 0.0), ("tukey", 0.1), ("tukey", 0.5), ("tukey", 1.0)])
def test_windows(handle, window):
    x = batch_sine_noise(4, 4096, 16, seed0=31)
    full_parity(handle, x, 16, 10, window=window)


@pytest.mark.parametrize("max_p", [0, 3, 14, 15, 30])
def test_max_rice_parameter(handle, max_p):
    """config::Prc::max_parameter; > 14 switches the written stream to RICE2 (bitrepr.rs:540-543)."""
    x = np.concatenate([batch_sine_noise(3, 4096, 16, seed0=3), batch_sine_noise(3, 4096, 24, seed0=4)])
    full_parity(handle, x, np.array([16] * 3 + [24] * 3, np.uint8), 10, max_p=max_p)


@pytest.mark.parametrize("bps", [8, 12, 16, 17, 20, 24, 25])
def test_bit_depths(handle, bps):
    """verify.rs:51-66 range 8..=24 (+1 for side channels); full-scale noise exercises the i64
    residual branch (lpc.rs:377-388) and, at 24/25 bits, f64 sums beyond 2^53."""
    x = np.stack([util.quantize(util.noise(100 + bps + k, 4096, 0.999), bps) for k in range(3)] +
                 [util.sine_noise(4096, bps, 50, 0.95, 0.04, seed=bps)])
    full_parity(handle, x, bps, 12)


def test_degenerate_signals(handle):
    """Digital silence (R == 0 -> all-zero coefficients, lpc.rs:647-657), DC, a single impulse,
    an impulse at t = 0 (windowed away: w[0] == 0), alternating extremes."""
    n = 4096
    x = np.zeros((7, n), np.int32)
    x[1, :] = 12345
    x[2, 1000] = 32767
    x[3, 0] = -32768
    x[4, ::2] = 32767
    x[4, 1::2] = -32768
    x[5, :] = -1
    x[6, : n // 2] = 1000
    # (near-)singular Toeplitz systems: coefficients have no meaningful fp tolerance across
    # summation orders, so only T0 (bit-exact vs canonical oracle), the R tolerance, the
    # integer stages and losslessness are asserted.
    gp, gres = full_parity(handle, x, 16, 10, coef_tolerance=False)
    assert int(gp["order"][0]) == 1 and (gp["coefs"][0] == 0).all() and int(gp["shift"][0]) == 15
    assert (gres[0] == 0).all()


def test_overflow_pattern_from_reference(handle):
    """src/lpc.rs:1415-1429 input (order 15, precision 13, rectangle) embedded in a 64-block."""
    sig = np.array([127] * 33 + [29] + [0] * 30, np.int32)
    full_parity(handle, sig[None, :], 8, 15, precision=13, window="rectangle")


def test_huge_residuals_saturate_like_reference(handle):
    """25-bit full-scale square-ish noise drives table entries toward MAX_P_TO_BITS = 2^27 - 1
    (rice.rs:51, 92-98) for small p; search + sum_quotients must still agree."""
    x = np.stack([util.quantize(util.noise(k, 16384, 1.0), 25) for k in range(2)])
    full_parity(handle, x, 25, 8, max_p=2)


def test_zero_subframes_and_bad_arguments(handle):
    cfg = gpu_cfg(10)
    p, r, _, _ = handle.qlpc_batch(np.zeros((0, 4096), np.int32), 16, cfg)
    assert p.shape == (0,) and r.shape == (0, 4096)
    with pytest.raises(_capi.FlacencHipError) as ei:
        handle.qlpc_batch(np.zeros((1, 32), np.int32), 16, cfg)  # < 64 never reaches the path
    assert ei.value.code == _capi.ERR_BAD_ARGUMENT
    bad = gpu_cfg(10)
    bad.quant_precision = 16
    with pytest.raises(_capi.FlacencHipError) as ei:
        handle.qlpc_batch(np.zeros((1, 4096), np.int32), 16, bad)
    assert ei.value.code == _capi.ERR_BAD_CONFIG


def test_large_batch_statistics(handle):
    """4096 subframes of BASELINE-config-2 audio: the unflagged (certified) integers are the reference
    order's on every subframe (round 5; until then: on >= 99.9 %), and every subframe is lossless."""
    ns = 4096
    x = batch_sine_noise(ns, 4096, 16, seed0=123456)
    gp, gres, _, _ = handle.qlpc_batch(x, 16, gpu_cfg(8))
    rp, rres, _, _ = orc.qlpc_batch(x, 16, orc_cfg(8), nthreads=8, want_fp=False)
    same = (gp["coefs"] == rp["coefs"]).all(axis=1) & (gp["shift"] == rp["shift"])
    print(f"\nidentical quantised coefficients vs reference order: {same.sum()}/{ns}")
    assert same.all()
    assert_records_equal(gp[same], rp[same])
    assert np.array_equal(gres[same], rres[same])
    cp, cres, _, _ = orc.qlpc_batch(x, 16, orc_cfg(8, acorr=orc.ACORR_CANONICAL), nthreads=8, want_fp=False)
    assert_records_equal(gp, cp)
    assert np.array_equal(gres, cres)
    check_lossless(x[::64], gp[::64], gres[::64])


# ------------------------------------------------------------------ stereo entry point ----
@pytest.mark.parametrize("n,bps,order", [(4096, 16, 8), (4096, 16, 10), (8192, 24, 24), (1152, 16, 12)])
def test_stereo_batch_equals_four_subframe_analyses(handle, n, bps, order):
    """flacenc_hip_stereo_qlpc_batch == estimated_qlpc on L, R, M = (l+r)>>1, S = l-r
    (coding.rs:476-491), S with bps + 1 (coding.rs:444); and M/S decode back to L/R
    (decode.rs:91-103)."""
    nf = 6
    frames = _capi.sigen_frames(nf, 2, n, bps, 200.0, 0.4, 0.1, seed=n + order)
    frames[:, 1] = (frames[:, 1] * 3) // 4 + frames[:, 0] // 8  # correlated channels
    params, residual = handle.stereo_qlpc_batch(frames, bps, gpu_cfg(order))
    ocfg = orc_cfg(order, acorr=orc.ACORR_CANONICAL)
    for f in range(nf):
        l, r = frames[f, 0], frames[f, 1]
        m, s = orc.stereo_to_midside(l, r)
        decoded = []
        for role, (sig, b) in enumerate(((l, bps), (r, bps), (m, bps), (s, bps + 1))):
            want = orc.estimated_qlpc(sig, b, ocfg)
            got = params[f, role]
            k = want["order"]
            assert int(got["status"]) == 0
            assert (int(got["order"]), int(got["shift"])) == (k, want["shift"])
            assert got["coefs"][:k].tolist() == want["coefs"].tolist()
            assert np.array_equal(residual[f, role], want["residual"])
            assert int(got["rice_order"]) == want["rice_order"]
            assert got["rice_params"][: 1 << want["rice_order"]].tolist() == want["rice_params"].tolist()
            assert int(got["code_bits"]) == want["code_bits"]
            assert int(got["subframe_bits"]) == want["subframe_bits"]
            decoded.append(orc.decode_lpc(sig[:k], got["coefs"][:k], int(got["shift"]), residual[f, role]))
        assert np.array_equal(decoded[0], l) and np.array_equal(decoded[1], r)
        l2, r2 = orc.midside_to_stereo(decoded[2], decoded[3])
        assert np.array_equal(l2, l) and np.array_equal(r2, r)


# ------------------------------------------------------------------ Rice search stress ----
@pytest.mark.parametrize("max_p", [30, 14, 6, 1])
def test_rice_search_window_stress(handle, max_p):
    """The 4096-block kernel searches only a provably sufficient window of Rice parameters
    (DESIGN.md 4.1).  Stress it with residual statistics that vary wildly inside a block:
    bursts, silence next to full scale, slow amplitude ramps, single outliers, per-partition
    alternation -- results must stay bit-identical to the exhaustive reference search."""
    n, rng = 4096, np.random.default_rng(20260 + max_p)
    sigs = []
    for k in range(96):
        kind = k % 8
        base = util.noise(5000 + k, n, 1.0)
        env = np.ones(n, np.float32)
        if kind == 0:    # burst in one 64-sample partition
            env[:] = 0.001
            q = int(rng.integers(0, 64)) * 64
            env[q:q + 64] = 0.9
        elif kind == 1:  # silence then loud
            env[: n // 2] = 0.0
            env[n // 2:] = 0.8
        elif kind == 2:  # exponential ramp over 5 orders of magnitude
            env = np.exp(np.linspace(np.log(1e-5), np.log(0.9), n)).astype(np.float32)
        elif kind == 3:  # alternating partitions quiet / loud
            env = np.where((np.arange(n) // 64) % 2 == 0, 0.002, 0.7).astype(np.float32)
        elif kind == 4:  # single outlier sample
            env[:] = 0.003
            env[int(rng.integers(100, n))] = 1.0
        elif kind == 5:  # tiny noise: residual of a few LSBs
            env[:] = 3.0 / 32768
        elif kind == 6:  # tone + modulated noise
            base = base * 0.2 + util.sine(n, 23.0 + k, 0.7)
            env = (0.05 + 0.9 * np.abs(util.sine(n, 700.0, 1.0))).astype(np.float32)
        else:            # random per-partition gains
            env = np.repeat(rng.choice([1e-4, 1e-3, 1e-2, 0.1, 0.9], size=64), 64).astype(np.float32)
        bps = 24 if k % 3 == 0 else 16
        sigs.append((util.quantize(base * env, bps), bps))
    x = np.stack([s for s, _ in sigs])
    bps = np.array([b for _, b in sigs], np.uint8)
    gp, gres, _, _ = handle.qlpc_batch(x, bps, gpu_cfg(8, max_p=max_p))
    cp, cres, _, _ = orc.qlpc_batch(x, bps, orc_cfg(8, max_p=max_p, acorr=orc.ACORR_CANONICAL), want_fp=False)
    assert_records_equal(gp, cp, f"max_p={max_p}")
    assert np.array_equal(gres, cres)
    check_lossless(x, gp, gres)


# ------------------------------------------------------------------ encode_frame on device ----
def _stereo_corpus(n=4096, bps=16):
    """Frames that exercise every channel assignment and every SubFrame kind."""
    fr = []
    base = _capi.sigen_frames(24, 2, n, bps, 200.0, 0.4, 0.05, seed=42)
    for f in range(24):
        l, r = base[f, 0].copy(), base[f, 1].copy()
        k = f % 8
        if k == 0:
            r = l.copy()                       # identical channels -> side is digital silence (Constant)
        elif k == 1:
            r = (l * 7) // 8                   # strongly correlated -> left/side or mid/side
        elif k == 2:
            l = (r * 3) // 4 + 5               # right/side territory
        elif k == 3:
            l[:] = 1234                        # constant left
        elif k == 4:
            l = util.quantize(util.noise(900 + f, n, 0.999), bps)  # white noise -> Verbatim
            r = util.quantize(util.noise(950 + f, n, 0.999), bps)
        elif k == 5:
            r = -l                             # anti-phase: mid ~ 0
        elif k == 6:
            l[:] = 0
            r[:] = 0                           # digital silence everywhere
        fr.append(np.stack([l, r]))
    return np.stack(fr).astype(np.int32)


@pytest.mark.parametrize("order,flags", [
    (8, dict()), (10, dict()), (12, dict(use_midside=False)),
    (8, dict(use_leftside=False, use_rightside=False)), (8, dict(use_constant=False)),
    (8, dict(use_lpc=False)), (8, dict(use_leftside=False, use_rightside=False, use_midside=False)),
])
def test_encode_stereo_frames_decision_equals_reference_controller(handle, order, flags):
    """flacenc_hip_encode_stereo_frames == encode_frame (coding.rs:530-544): encode_subframe's choice
    per candidate (coding.rs:384-418, use_fixed = false) and try_stereo_coding's assignment
    (coding.rs:493-522, strict '<' in the order LeftSide, RightSide, MidSide), restated by the oracle."""
    bps = 16
    x = _stereo_corpus()
    cfg = _capi.make_frame_config(gpu_cfg(order), **flags)
    got, gres = handle.encode_stereo_frames(x, bps, cfg)
    want, wres = orc.encode_stereo_frames(x, bps, orc_cfg(order, acorr=orc.ACORR_CANONICAL), **flags)
    for f in range(x.shape[0]):
        g, w = got[f], want[f]
        assert int(g["channel_assignment"]) == int(w["channel_assignment"]), f
        assert g["role"].tolist() == w["role"].tolist(), f
        assert g["kind"].tolist() == w["kind"].tolist(), f
        assert g["dc_offset"].tolist() == w["dc_offset"].tolist(), f
        assert g["bits"].tolist() == w["bits"].tolist(), f
        for c in range(2):
            if int(g["kind"][c]) == 3:
                gl, wl = g["lpc"][c], w["lpc"][c]
                for fld in ("order", "shift", "precision", "rice_order", "status", "code_bits",
                            "subframe_bits", "sum_quotients"):
                    assert int(gl[fld]) == int(wl[fld]), (f, c, fld)
                assert gl["coefs"].tolist() == wl["coefs"].tolist()
                assert gl["rice_params"].tolist() == wl["rice_params"].tolist()
            assert np.array_equal(gres[f, c], wres[f, c]), (f, c)
    kinds = set(got["kind"].ravel().tolist())
    assigns = set(got["channel_assignment"].tolist())
    if not flags:
        assert kinds == {0, 1, 3} and len(assigns) >= 3, (kinds, assigns)
    # losslessness of the chosen representation (decode.rs:61-113)
    for f in range(x.shape[0]):
        g = got[f]
        ch = []
        for c in range(2):
            role = int(g["role"][c])
            l, r = x[f, 0], x[f, 1]
            sig = [l, r, *orc.stereo_to_midside(l, r)][role]
            if int(g["kind"][c]) == 0:
                ch.append(np.full(4096, int(g["dc_offset"][c]), np.int32))
            elif int(g["kind"][c]) == 1:
                ch.append(sig.copy())
            else:
                p = g["lpc"][c]
                k = int(p["order"])
                ch.append(orc.decode_lpc(sig[:k], p["coefs"][:k], int(p["shift"]), gres[f, c]))
        a = int(g["channel_assignment"])
        if a == 1:
            ch[1] = ch[0] - ch[1]
        elif a == 2:
            ch[0] = ch[0] + ch[1]
        elif a == 3:
            ch[0], ch[1] = orc.midside_to_stereo(ch[0], ch[1])
        assert np.array_equal(ch[0], x[f, 0]) and np.array_equal(ch[1], x[f, 1]), f


def _fixed_corpus(n=4096, bps=16):
    """_stereo_corpus plus frames on which the fixed-LPC candidate beats the QLPC one (smooth
    ramps / parabolas / steps / square waves / quiet 8-bit-like tones) and loud material on
    which the estimator's sums exceed 2^24."""
    t = np.arange(n)
    extra = [
        np.stack([(t // 7), (t // 5) + 3]),
        np.stack([(t - 2048) ** 2 // 400, (t - 1000) ** 2 // 300 - 2000]),
        np.stack([np.repeat(np.arange(64) * 10, 64), np.repeat(np.arange(32) * 25, 128)]),
        np.stack([(t // 50 % 2) * 200 - 100, (t // 80 % 2) * 300 - 150]),
        np.stack([np.abs((t % 200) - 100) * 50, np.abs((t % 300) - 150) * 40]),
        np.stack([util.quantize(util.sine(n, 100, 0.6), 8), util.quantize(util.sine(n, 70, 0.5, phase=1.0), 8)]),
        np.stack([util.quantize(util.noise(7, n, 0.999), bps), util.quantize(util.sine(n, 3.1, 0.99), bps)]),
        np.stack([util.sine_noise(n, bps, 2.3, 0.9, 0.09, seed=8), util.sine_noise(n, bps, 2.7, 0.9, 0.09, seed=9)]),
    ]
    return np.concatenate([_stereo_corpus(n, bps), np.stack(extra).astype(np.int32)])


def _check_frames_against_oracle(x, bps, got, gres, want, wres):
    for f in range(x.shape[0]):
        g, w = got[f], want[f]
        for fld in ("channel_assignment", "role", "kind", "dc_offset", "bits"):
            assert g[fld].tolist() == w[fld].tolist(), (f, fld, g[fld].tolist(), w[fld].tolist())
        for c in range(2):
            if int(g["kind"][c]) >= 2:
                gl, wl = g["lpc"][c], w["lpc"][c]
                for fld in ("order", "shift", "precision", "rice_order", "status", "code_bits",
                            "subframe_bits", "sum_quotients"):
                    assert int(gl[fld]) == int(wl[fld]), (f, c, fld)
                assert gl["coefs"].tolist() == wl["coefs"].tolist()
                assert gl["rice_params"].tolist() == wl["rice_params"].tolist()
            assert np.array_equal(gres[f, c], wres[f, c]), (f, c)


def _decode_frames(x, got, gres):
    """Decode for Frame / SubFrame (decode.rs:61-113, 159-217): Constant, Verbatim, FixedLpc, Lpc."""
    n = x.shape[2]
    for f in range(x.shape[0]):
        g = got[f]
        ch = []
        for c in range(2):
            role = int(g["role"][c])
            l, r = x[f, 0], x[f, 1]
            sig = [l, r, *orc.stereo_to_midside(l, r)][role]
            kind = int(g["kind"][c])
            if kind == 0:
                ch.append(np.full(n, int(g["dc_offset"][c]), np.int32))
            elif kind == 1:
                ch.append(sig.copy())
            else:
                p = g["lpc"][c]
                k = int(p["order"])
                if kind == 2:
                    assert p["coefs"][:4].tolist() == orc.FIXED_LPC_COEFS[k] and int(p["shift"]) == 0
                ch.append(orc.decode_lpc(sig[:k], p["coefs"][:k], int(p["shift"]), gres[f, c]))
        a = int(g["channel_assignment"])
        if a == 1:
            ch[1] = ch[0] - ch[1]
        elif a == 2:
            ch[0] = ch[0] + ch[1]
        elif a == 3:
            ch[0], ch[1] = orc.midside_to_stereo(ch[0], ch[1])
        assert np.array_equal(ch[0], x[f, 0]) and np.array_equal(ch[1], x[f, 1]), f


@pytest.mark.parametrize("order,fixed", [
    (8, dict()), (10, dict(fixed_max_order=2)), (12, dict(fixed_partitions=64)),
    (8, dict(fixed_partitions=1)), (8, dict(fixed_partitions=4, fixed_max_order=3)),
    (8, dict(fixed_order_sel=0)), (10, dict(fixed_order_sel=0, fixed_max_order=1)),
    (8, dict(fixed_max_order=0)),
])
def test_encode_stereo_frames_with_fixed_lpc_candidate(handle, order, fixed):
    """The reference's default SubFrameCoding (use_fixed = true): fixed_lpc (coding.rs:298-331) with
    either order selector, encode_subframe's three-way choice (coding.rs:384-418) and
    try_stereo_coding, all on the GPU == the oracle with the canonical summation definitions."""
    bps = 16
    x = _fixed_corpus()
    cfg = _capi.make_frame_config(gpu_cfg(order), use_fixed=True, **fixed)
    got, gres = handle.encode_stereo_frames(x, bps, cfg)
    ofixed = orc.make_fixed_config(max_order=fixed.get("fixed_max_order", 4),
                                   order_sel=fixed.get("fixed_order_sel", 1),
                                   partitions=fixed.get("fixed_partitions", 16),
                                   sum_mode=orc.SUMABS_CANONICAL)
    ocfg = orc.make_frame_config(orc_cfg(order, acorr=orc.ACORR_CANONICAL), use_fixed=True, fixed=ofixed)
    want, wres = orc.encode_stereo_frames_cfg(x, bps, ocfg)
    _check_frames_against_oracle(x, bps, got, gres, want, wres)
    if not fixed:
        assert set(got["kind"].ravel().tolist()) == {0, 1, 2, 3}
    _decode_frames(x, got, gres)


def test_fixed_lpc_selector_keys_equal_oracle(hooks_handle):
    """The selector's per-order keys (estimate_entropy + bps*order, coding.rs:271; BitCount bits,
    coding.rs:249) for L, R, M, S of every frame, bit for bit; and against the reference's own two
    summation orders wherever every partition sum stays below 2^24."""
    handle = hooks_handle  # (debug_set_fixed_keys: the hooks build, same kernels)
    import torch
    bps = 16
    x = _fixed_corpus()
    F = x.shape[0]
    for sel in (1, 0):
        keys = torch.zeros((F * 4, 8), dtype=torch.int64, device="cuda")
        handle.debug_set_fixed_keys(keys.data_ptr())
        try:
            handle.encode_stereo_frames(x, bps, _capi.make_frame_config(gpu_cfg(8), use_fixed=True,
                                                                         fixed_order_sel=sel))
        finally:
            handle.debug_set_fixed_keys(0)
        k = keys.cpu().numpy().astype(np.uint64).reshape(F, 4, 8)
        exact_everywhere = 0
        for f in range(F):
            l, r = x[f, 0], x[f, 1]
            for role, sig in enumerate([l, r, *orc.stereo_to_midside(l, r)]):
                b = bps + (1 if role == 3 else 0)
                fc = orc.make_fixed_config(order_sel=sel, sum_mode=orc.SUMABS_CANONICAL)
                want = orc.fixed_lpc(sig, b, 2 ** 63, fc)["estimate"]
                assert k[f, role, :5].tolist() == want, (sel, f, role)
                if sel == 1:
                    errs = orc.reset_fixed_lpc_errors(sig)
                    small = all(np.abs(errs[o].astype(np.int64)).reshape(16, -1).sum(axis=1).max() < 2 ** 24
                                for o in range(5))
                    if small:
                        exact_everywhere += 1
                        for mode in (orc.SUMABS_STABLE, orc.SUMABS_NIGHTLY):
                            ref = orc.fixed_lpc(sig, b, 2 ** 63, orc.make_fixed_config(sum_mode=mode))
                            assert ref["estimate"] == want, (f, role, mode)
        if sel == 1:
            assert exact_everywhere >= 2 * F  # most of the corpus is in the exactly-pinned regime


def test_encode_stereo_frames_fixed_24bit(handle):
    """24-bit material: the side channel has 25 bits, order-4 differences reach 2^28 and the
    estimator's partition sums 2^36 -- the exact-integer sums must still agree."""
    bps = 24
    x = _capi.sigen_frames(6, 2, 4096, bps, 37.0, 0.7, 0.2, seed=77)
    x[1] = x[1] // 4096
    x[2, 1] = -x[2, 0]
    x[3] = (np.arange(4096)[None, :] * np.array([[1000], [-900]])).astype(np.int32)
    x[4, 0] = np.where(np.arange(4096) % 2 == 0, 2 ** 23 - 1, -2 ** 23)   # worst-case alternation
    x[4, 1] = -x[4, 0] - 1
    cfg = _capi.make_frame_config(gpu_cfg(8), use_fixed=True)
    got, gres = handle.encode_stereo_frames(x, bps, cfg)
    ocfg = orc.make_frame_config(orc_cfg(8, acorr=orc.ACORR_CANONICAL), use_fixed=True,
                                 fixed=orc.make_fixed_config(sum_mode=orc.SUMABS_CANONICAL))
    want, wres = orc.encode_stereo_frames_cfg(x, bps, ocfg)
    _check_frames_against_oracle(x, bps, got, gres, want, wres)
    _decode_frames(x, got, gres)


@pytest.mark.parametrize("n,bps,sel,parts,max_order", [
    (4096, 16, 1, 16, 4), (4608, 16, 1, 16, 4), (1152, 16, 1, 7, 4), (8192, 24, 1, 64, 4),
    (576, 16, 1, 1, 2), (100, 8, 1, 64, 4), (16384, 24, 1, 16, 4), (20000, 16, 1, 33, 3),
    (4096, 16, 0, 16, 4), (4608, 16, 0, 16, 2), (192, 16, 0, 16, 4), (8192, 24, 0, 16, 4),
    # the big-block selection kernel's group sizes: one lane, all 64 lanes of a pass, a partition per pass;
    # and a partition longer than a pass (generic kernel)
    (4096, 16, 1, 64, 3), (16384, 24, 1, 4, 4), (8192, 16, 1, 2, 1), (8192, 24, 1, 1, 4), (4096, 24, 1, 8, 0),
])
def test_fixed_lpc_batch_any_shape(handle, n, bps, sel, parts, max_order):
    """flacenc_hip_fixed_lpc_batch == fixed_lpc (coding.rs:298-331) on ragged / large / tiny blocks,
    any ApproxEnt partition count (incl. partitions shorter than the warm-up and empty ones) and
    BitCount: selector keys, chosen order, Rice partition, bit counts and the error signal."""
    sigs = [util.sine_noise(n, bps, 200, 0.4, 0.05, seed=1), util.sine_noise(n, bps, 31, 0.7, 0.3, seed=2),
            util.quantize(util.sine(n, 100, 0.6), bps), (np.arange(n) // 7).astype(np.int32),
            ((np.arange(n) - n // 2) ** 2 // 400 % (1 << (bps - 2))).astype(np.int32),
            np.full(n, 77, np.int32), np.zeros(n, np.int32),
            util.quantize(util.noise(5, n, 0.999), bps),
            np.where(np.arange(n) % 2 == 0, 2 ** (bps - 1) - 1, -2 ** (bps - 1)).astype(np.int32)]
    x = np.stack(sigs).astype(np.int32)
    bpsv = np.full(len(sigs), bps, np.uint8)
    bpsv[1] = bps + 1 if bps < 25 else bps
    cfg = _capi.make_frame_config(gpu_cfg(8), use_fixed=True, fixed_order_sel=sel, fixed_partitions=parts,
                                  fixed_max_order=max_order)
    params, resid, keys = handle.fixed_lpc_batch(x, bpsv, cfg)
    fc = orc.make_fixed_config(max_order=max_order, order_sel=sel, partitions=parts,
                               sum_mode=orc.SUMABS_CANONICAL)
    for k in range(len(sigs)):
        w = orc.fixed_lpc(x[k], int(bpsv[k]), 2 ** 63, fc)
        p = params[k]
        assert int(p["order"]) == w["order"], k
        assert int(keys[k]) == w["estimate"][w["order"]], k
        for fld in ("rice_order", "code_bits", "subframe_bits", "sum_quotients"):
            assert int(p[fld]) == int(w[fld]), (k, fld)
        assert int(p["shift"]) == 0 and int(p["precision"]) == 0 and int(p["status"]) == 0
        assert p["coefs"][:4].tolist() == orc.FIXED_LPC_COEFS[w["order"]]
        assert p["rice_params"][: 1 << w["rice_order"]].tolist() == w["rice_params"].tolist()
        assert np.array_equal(resid[k], w["residual"]), k
        assert np.array_equal(orc.decode_fixed(x[k][: w["order"]], resid[k]), x[k]), k


@pytest.mark.parametrize("n,sel", [(4608, 1), (4096, 1), (2048, 0)])
def test_fixed_lpc_batch_stereo_roles(handle, n, sel):
    """STEREO_FRAMES layout: L, R and the M = (l+r)>>1, S = l-r formed on the GPU (coding.rs:476-484),
    side channel with one more bit per sample (coding.rs:444)."""
    bps = 16
    x = _capi.sigen_frames(5, 2, n, bps, 90.0, 0.5, 0.02, seed=31)
    x[1, 1] = x[1, 0] // 2 + 3
    x[2] = (np.arange(n)[None, :] * np.array([[3], [-2]]) // 5).astype(np.int32)
    cfg = _capi.make_frame_config(gpu_cfg(8), use_fixed=True, fixed_order_sel=sel)
    params, resid, keys = handle.fixed_lpc_batch(x, bps, cfg, stereo=True)
    fc = orc.make_fixed_config(order_sel=sel, sum_mode=orc.SUMABS_CANONICAL)
    for f in range(x.shape[0]):
        l, r = x[f, 0], x[f, 1]
        for role, sig in enumerate([l, r, *orc.stereo_to_midside(l, r)]):
            w = orc.fixed_lpc(sig, bps + (1 if role == 3 else 0), 2 ** 63, fc)
            p = params[f, role]
            assert int(p["order"]) == w["order"] and int(keys[f, role]) == w["estimate"][w["order"]], (f, role)
            assert int(p["subframe_bits"]) == w["subframe_bits"] and int(p["code_bits"]) == w["code_bits"]
            assert np.array_equal(resid[f, role], w["residual"]), (f, role)


@pytest.mark.parametrize("bps,rate,first,step,use_fixed", [
    (16, 44100, 0, 1, True), (16, 48000, 120, 1, False), (16, 12345, 2 ** 20 + 5, 8, True),
    (24, 96000, 2 ** 26, 3, True), (16, 17000, 2 ** 30, 1, True), (16, 1234567, 2047, 1, True),
])
def test_pack_stereo_frames_bytes_equal_reference_writer(handle, bps, rate, first, step, use_fixed):
    """flacenc_hip_pack_stereo_frames == Frame::write (bitrepr.rs:289-319) restated by the oracle,
    byte for byte, for every SubFrame kind / channel assignment / header variant (frame-number
    UTF-8 lengths 1..6, coded and explicit sample rates); and the bytes decode to the input with
    the independent parser of tests/flac_parse.py (sync code, CRC-8, CRC-16, Rice coding)."""
    import flac_parse
    x = _fixed_corpus()
    if bps == 24:
        x = (x.astype(np.int64) * 181).astype(np.int32)   # 24-bit range, same structure
        x[4] = np.stack([util.quantize(util.noise(21, 4096, 0.999), 24), util.quantize(util.noise(22, 4096, 0.999), 24)])
    cfg = _capi.make_frame_config(gpu_cfg(8), use_fixed=use_fixed)
    res, resid = handle.encode_stereo_frames(x, bps, cfg)
    frames = handle.pack_stereo_frames(x, res, resid, bps, rate, first, step)
    kinds = set()
    for f in range(x.shape[0]):
        want = orc.write_stereo_frame(res[f], x[f, 0], x[f, 1], bps, rate, first + f * step, resid[f, 0], resid[f, 1])
        assert frames[f] == want, (f, len(frames[f]), len(want))
        sub_bits = sum(int(res[f]["bits"][r]) for r in res[f]["role"])
        hdr = orc.write_frame_header(4096, 1, bps, rate, False, first + f * step)
        assert len(frames[f]) * 8 == (8 * len(hdr) + sub_bits + 7) // 8 * 8 + 16   # Frame::count_bits
        if f % 3 == 0 or f == 4 or f >= 24:
            got = flac_parse.parse_frame(frames[f], stream_bps=bps, stream_rate=rate)
            assert got["number"] == first + f * step and got["length"] == len(frames[f])
            assert np.array_equal(got["channels"], x[f]), f
            kinds.update(got["kinds"])
    assert kinds >= ({"constant", "verbatim", "fixed", "lpc"} if use_fixed else {"constant", "verbatim", "lpc"})


def test_pack_stereo_frames_device_pipeline_checksum(handle):
    """encode + pack entirely on the device for a large batch: every frame's CRC-16 (computed by the
    independent parser's CRC over the bytes) matches its footer, lengths equal Frame::count_bits,
    and a sample of frames decodes back to the input."""
    import torch
    import flac_parse
    F, n, bps = 512, 4096, 16
    host = _capi.sigen_frames(F, 2, n, bps, 200.0, 0.4, 0.4, seed=99)
    host[7] = host[7] // 64
    x = torch.from_numpy(host).cuda()
    res = torch.empty((F, 752), dtype=torch.uint8, device="cuda")
    resid = torch.empty((F * 2, n), dtype=torch.int32, device="cuda")
    stride = handle.frame_bytes_bound(n, bps)
    out = torch.zeros((F, stride), dtype=torch.uint8, device="cuda")
    lens = torch.zeros(F, dtype=torch.int32, device="cuda")
    cfg = _capi.make_frame_config(gpu_cfg(8), use_fixed=True)
    s = torch.cuda.current_stream().cuda_stream
    handle.encode_stereo_frames_device(cfg, x.data_ptr(), F, n, n, bps, res.data_ptr(), resid.data_ptr(), n, stream=s)
    handle.pack_stereo_frames_device(x.data_ptr(), F, n, n, res.data_ptr(), resid.data_ptr(), n, bps, 44100, 0, 1,
                                     out.data_ptr(), stride, lens.data_ptr(), stream=s)
    lens2 = torch.zeros(F, dtype=torch.int32, device="cuda")
    handle.stereo_frame_lengths_device(res.data_ptr(), F, n, bps, 44100, 0, 1, lens2.data_ptr(), stream=s)
    torch.cuda.synchronize()
    assert torch.equal(lens, lens2)   # Frame::count_bits from the records alone == bytes really packed
    o, ln = out.cpu().numpy(), lens.cpu().numpy()
    r = np.frombuffer(res.cpu().numpy().tobytes(), dtype=_capi.FRAME_RESULT_DTYPE)
    for f in range(F):
        b = bytes(o[f, :ln[f]])
        assert b[:2] == bytes([0xFF, 0xF8])
        assert flac_parse.crc16(b[:-2]) == (b[-2] << 8 | b[-1]), f
        hdr_len = 6 if f < 128 else 7
        assert flac_parse.crc8(b[:hdr_len - 1]) == b[hdr_len - 1], f
        sub_bits = sum(int(r[f]["bits"][k]) for k in r[f]["role"])
        assert ln[f] * 8 == (8 * hdr_len + sub_bits + 7) // 8 * 8 + 16
    for f in (0, 7, 127, 128, 511):
        got = flac_parse.parse_frame(bytes(o[f, :ln[f]]))
        assert got["number"] == f and np.array_equal(got["channels"], host[f])


@pytest.mark.parametrize("n,bps,order,odd_stride", [
    (4096, 24, 8, False), (8192, 24, 24, False), (12288, 24, 12, True), (16384, 24, 32, False),
    (20480, 16, 16, False), (8192, 16, 10, True), (4096, 16, 8, True),
])
def test_packer_aligned_runs_all_rounds(handle, n, bps, order, odd_stride):
    """The packer's run-of-16 path (block sizes that are multiples of 4096: one to five rounds of 4096
    samples per subframe, partitions of 16 k samples) against the oracle's Frame::write: warm-up runs of 8,
    24 and 32 samples (the first two threads emit nothing or a part), RICE2 (loud 24-bit noise: parameters
    above 14), small residuals next to huge ones, constant and verbatim subframes, and residual rows that
    are not 16-byte aligned (device entry point with an odd row stride: scalar loads)."""
    import torch
    F = 5
    x = _capi.sigen_frames(F, 2, n, bps, 90.0, 0.45, 0.02, seed=31 * n + order, nthreads=2)
    x[1] = np.stack([util.quantize(util.noise(5, n, 0.999), bps), util.quantize(util.noise(6, n, 0.999), bps)])
    x[2, 0] = -3
    x[3, :, n // 2:] //= 4096  # quiet second half: parameters differ wildly between partitions
    x[4] = np.stack([util.quantize(util.noise(7, n, 0.999), bps), x[0, 1]])
    cfg = _capi.make_frame_config(gpu_cfg(order), use_fixed=True)
    res, resid = handle.encode_stereo_frames(x, bps, cfg)
    want = [orc.write_stereo_frame(res[f], x[f, 0], x[f, 1], bps, 96000, 1000 + f, resid[f, 0], resid[f, 1])
            for f in range(F)]
    if not odd_stride:
        got = handle.pack_stereo_frames(x, res, resid, bps, 96000, 1000, 1)
    else:
        rs = n + 1
        dres = torch.zeros((F * 2, rs), dtype=torch.int32, device="cuda")
        dres[:, :n] = torch.from_numpy(np.ascontiguousarray(resid.reshape(F * 2, n))).cuda()
        dx = torch.from_numpy(x).cuda()
        drec = torch.from_numpy(np.frombuffer(res.tobytes(), np.uint8).copy()).cuda()
        stride = (handle.frame_bytes_bound(n, bps) + 15) // 16 * 16
        out = torch.zeros((F, stride), dtype=torch.uint8, device="cuda")
        lens = torch.zeros(F, dtype=torch.int32, device="cuda")
        handle.pack_stereo_frames_device(dx.data_ptr(), F, n, n, drec.data_ptr(), dres.data_ptr(), rs, bps, 96000, 1000, 1,
                                         out.data_ptr(), stride, lens.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        o, ln = out.cpu().numpy(), lens.cpu().numpy()
        got = [bytes(o[f, :ln[f]]) for f in range(F)]
    for f in range(F):
        assert got[f] == want[f], (f, len(got[f]), len(want[f]))
    rice2 = 0
    for f in range(F):
        for c in range(2):
            if int(res[f]["kind"][c]) >= 2:
                k = 1 << int(res[f]["lpc"][c]["rice_order"])
                rice2 += int((res[f]["lpc"][c]["rice_params"][:k] > 14).any())
    if bps == 24:
        assert rice2 > 0


def test_flac_stream_end_to_end(handle):
    """tools/encode_flac.py: "fLaC" + STREAMINFO + GPU-packed frames.  Every frame parses with the
    independent parser, frame numbers run 0..F-1, STREAMINFO carries the right geometry and the MD5
    of the interleaved input (src/source.rs:406-428), and the decoded audio is the input."""
    import hashlib
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import encode_flac
    import flac_parse
    bps, rate, n = 16, 44100, 4096
    frames = _capi.sigen_frames(9, 2, n, bps, rate / 440.0, 0.8, 0.2, seed=1)   # flacenc-bin's test tone
    frames[3, 1] = frames[3, 0]
    frames[5] //= 300
    data, res = encode_flac.encode(frames, bps, rate, handle)
    assert data[:4] == b"fLaC" and data[4] == 0x80 and int.from_bytes(data[5:8], "big") == 34
    si = data[8:42]
    assert int.from_bytes(si[0:2], "big") == n and int.from_bytes(si[2:4], "big") == n
    packed = int.from_bytes(si[10:18], "big")
    assert packed >> 44 == rate and (packed >> 41) & 7 == 1 and (packed >> 36) & 31 == bps - 1
    assert packed & ((1 << 36) - 1) == 9 * n
    inter = np.ascontiguousarray(frames.transpose(0, 2, 1)).reshape(-1).astype("<i2")
    assert si[18:34] == hashlib.md5(inter.tobytes()).digest()
    pos, sizes = 42, []
    for f in range(9):
        got = flac_parse.parse_frame(data[pos:])
        assert got["number"] == f and np.array_equal(got["channels"], frames[f]), f
        sizes.append(got["length"])
        pos += got["length"]
    assert pos == len(data)
    assert int.from_bytes(si[4:7], "big") == min(sizes) and int.from_bytes(si[7:10], "big") == max(sizes)
    # arbitrary length: three whole blocks and a 1234-sample tail block (a second, one-frame batch)
    pcm = np.ascontiguousarray(frames[:4].transpose(0, 2, 1)).reshape(-1, 2)[: 3 * n + 1234]
    data, res = encode_flac.encode_pcm(pcm, bps, rate, handle)
    assert len(res) == 4 and int.from_bytes(data[8:42][10:18], "big") & ((1 << 36) - 1) == len(pcm)
    pos, t = 42, 0
    for f in range(4):
        got = flac_parse.parse_frame(data[pos:])
        assert got["number"] == f and got["block_size"] == (n if f < 3 else 1234)
        assert np.array_equal(got["channels"], pcm[t:t + got["block_size"]].T), f
        pos, t = pos + got["length"], t + got["block_size"]
    assert pos == len(data) and t == len(pcm)


@pytest.mark.parametrize("n,bps,order,kw", [
    (4608, 16, 8, dict()), (1152, 16, 10, dict()), (576, 16, 12, dict(fixed_partitions=12)),
    (8192, 24, 24, dict()), (192, 16, 8, dict(fixed_order_sel=0)), (4096, 16, 16, dict()),
    (4096, 16, 8, dict(fixed_partitions=12)), (16384, 24, 12, dict(use_lpc=False)),
    (2304, 16, 8, dict(use_fixed=False)), (100, 8, 4, dict()),
    # the selector's 18-samples-per-lane sums (blocks of 2^k * 18 samples with partitions of whole lanes) at the
    # widest inputs -- frame 5: +-full scale alternating, the side channel 25 bits of it -- and next to partition
    # counts that do not qualify (1152 / 12 = 96 is not a multiple of 18; 1152 / 128 = 9)
    (1152, 24, 8, dict()), (2304, 24, 10, dict()), (576, 24, 6, dict()), (2304, 16, 8, dict(fixed_partitions=8)),
    (1152, 24, 8, dict(fixed_partitions=64)), (1152, 16, 8, dict(fixed_partitions=32)), (1152, 24, 8, dict(fixed_partitions=12)),
    (18432, 24, 8, dict()),
    # ... and the same sums on whole 16-sample rows (power-of-two blocks), with a count that does not qualify beside them
    (2048, 24, 8, dict()), (256, 24, 6, dict()), (512, 24, 4, dict()), (1024, 16, 10, dict(fixed_partitions=64)),
    (1024, 24, 8, dict(fixed_partitions=12)),
])
def test_encode_stereo_frames_any_shape(handle, n, bps, order, kw):
    """flacenc_hip_encode_stereo_frames outside the fused kernel's shape (ragged / large / tiny
    blocks, orders > 12, partition counts that are not lane groups): candidate batches + the
    controller kernel == the oracle's encode_frame, and the frames pack and parse back."""
    import flac_parse
    base = _capi.sigen_frames(6, 2, n, bps, 120.0, 0.5, 0.03, seed=n + order)
    t = np.arange(n)
    base[1, 1] = base[1, 0]
    half = 1 << (bps - 2)
    base[2] = np.stack([(t // 7) % half, (t * t // (n // 4 + 1)) % half - half // 2]).astype(np.int32)
    base[3, 0] = 77
    base[4] = np.stack([util.quantize(util.noise(1, n, 0.999), bps), util.quantize(util.noise(2, n, 0.999), bps)])
    alt = np.where(t % 2 == 0, (1 << (bps - 1)) - 1, -(1 << (bps - 1))).astype(np.int32)
    base[5] = np.stack([alt, -1 - alt])  # fourth differences of 16 x full scale; side = 2 x full scale + 1
    use_fixed = kw.pop("use_fixed", True)
    cfg = _capi.make_frame_config(gpu_cfg(order), use_fixed=use_fixed, **kw)
    got, gres = handle.encode_stereo_frames(base, bps, cfg)
    ofixed = orc.make_fixed_config(max_order=kw.get("fixed_max_order", 4), order_sel=kw.get("fixed_order_sel", 1),
                                   partitions=kw.get("fixed_partitions", 16), sum_mode=orc.SUMABS_CANONICAL)
    ocfg = orc.make_frame_config(orc_cfg(order, acorr=orc.ACORR_CANONICAL), use_fixed=use_fixed,
                                 use_lpc=kw.get("use_lpc", True), fixed=ofixed)
    want, wres = orc.encode_stereo_frames_cfg(base, bps, ocfg)
    _check_frames_against_oracle(base, bps, got, gres, want, wres)
    frames = handle.pack_stereo_frames(base, got, gres, bps, 48000, 5, 1)
    for f in range(base.shape[0]):
        assert frames[f] == orc.write_stereo_frame(got[f], base[f, 0], base[f, 1], bps, 48000, 5 + f, gres[f, 0], gres[f, 1])
        if n <= 4608:
            parsed = flac_parse.parse_frame(frames[f], stream_bps=bps)
            assert parsed["block_size"] == n and np.array_equal(parsed["channels"], base[f]), f


@pytest.mark.parametrize("channels,n,bps,order,use_fixed,kw", [
    (1, 4096, 16, 8, True, {}), (8, 4096, 16, 10, True, {}), (3, 1152, 16, 12, True, {}),
    (5, 4608, 24, 8, False, {}), (1, 300, 8, 4, True, {}), (2, 4096, 16, 12, False, {}),
    (3, 4096, 24, 8, True, dict(fixed_order_sel=0)), (2, 4096, 16, 8, True, dict(fixed_partitions=12)),
    (4, 4096, 16, 10, True, dict(fixed_partitions=64, fixed_max_order=2)),
])
def test_encode_and_pack_independent_channel_frames(handle, channels, n, bps, order, use_fixed, kw):
    """Mono / multi-channel streams (encode_frame with Independent(n), coding.rs:537-541):
    flacenc_hip_encode_frames == encode_subframe per channel, flacenc_hip_pack_frames == Frame::write,
    both against the oracle, and the bytes parse back to the input (BASELINE config 4 is the 8-channel case)."""
    import flac_parse
    x = _capi.sigen_frames(4, channels, n, bps, 150.0, 0.5, 0.04, seed=channels * 1000 + n)
    half = 1 << (bps - 2)
    x[1, 0] = (np.arange(n) // 9) % half          # FixedLpc territory
    x[2, channels - 1] = -5                        # Constant
    x[3, 0] = util.quantize(util.noise(4, n, 0.999), bps)   # Verbatim
    cfg = _capi.make_frame_config(gpu_cfg(order), use_fixed=use_fixed, **kw)
    res, resid = handle.encode_frames(x, bps, cfg)
    ocfg = orc.make_frame_config(orc_cfg(order, acorr=orc.ACORR_CANONICAL), use_fixed=use_fixed,
                                 fixed=orc.make_fixed_config(max_order=kw.get("fixed_max_order", 4),
                                                             order_sel=kw.get("fixed_order_sel", 1),
                                                             partitions=kw.get("fixed_partitions", 16),
                                                             sum_mode=orc.SUMABS_CANONICAL))
    packed = handle.pack_frames(x, res, resid, bps, 44100, 70, 1)
    kinds = set()
    for f in range(x.shape[0]):
        subs = []
        for c in range(channels):
            w = orc.encode_subframe(x[f, c], bps, ocfg)
            g = res[f, c]
            assert int(g["kind"]) == w["kind"] and int(g["bits"]) == w["bits"], (f, c)
            kinds.add(w["kind"])
            p = g["params"]
            if w["kind"] >= 2:
                src = w["lpc"] if w["kind"] == 3 else w["fixed"]
                order_w = int(src.qp.order) if w["kind"] == 3 else int(src.order)
                assert int(p["order"]) == order_w and int(p["rice_order"]) == int(src.rice_order)
                assert int(p["subframe_bits"]) == int(src.subframe_bits) == w["bits"]
                assert np.array_equal(resid[f, c], w["residual"]), (f, c)
            else:
                assert not resid[f, c].any()
            subs.append(dict(kind=int(g["kind"]), bps=bps, samples=x[f, c], dc_offset=int(g["dc_offset"]),
                             order=int(p["order"]), shift=int(p["shift"]), precision=int(p["precision"]),
                             coefs=p["coefs"], rice_order=int(p["rice_order"]), rice_params=p["rice_params"],
                             residual=resid[f, c]))
        want = orc.write_frame(n, 0, bps, 44100, 70 + f, subs)
        assert packed[f] == want, (f, len(packed[f]), len(want))
        got = flac_parse.parse_frame(packed[f], stream_bps=bps)
        assert got["channel_tag"] == channels - 1 and got["number"] == 70 + f
        assert np.array_equal(got["channels"], x[f]), f
    if n >= 1152:
        assert {0, 1, 3} <= kinds and (2 in kinds or not use_fixed)


@pytest.mark.parametrize("order,bps,use_fixed", [(8, 16, True), (12, 16, False), (10, 24, True)])
def test_frame_pipeline_soak(handle, order, bps, use_fixed):
    """768 frames of mixed material (tones with little or much noise, near-silence, hard-panned and
    anti-phase channels, full-scale bursts) through analysis -> decision -> packing, every record,
    residual row and packed byte against the oracle."""
    rng = np.random.default_rng(order * 100 + bps)
    parts = []
    for k in range(12):
        period = float(rng.uniform(2.5, 900.0))
        amp = float(rng.choice([0.001, 0.02, 0.3, 0.9]))
        namp = float(rng.choice([0.0, 0.0005, 0.01, 0.2, 0.7])) * min(1.0, (0.99 - amp) / 0.7 + 0.3)
        namp = min(namp, 0.99 - amp)
        parts.append(_capi.sigen_frames(64, 2, 4096, bps, period, amp, max(namp, 0.0), seed=1000 * order + k))
    x = np.concatenate(parts)
    x[5::37, 1] = x[5::37, 0]
    x[9::41, 1] = -x[9::41, 0]
    x[11::53, 0] = 0
    x[13::59] = x[13::59] // 1024
    x[17::61, 1] = x[17::61, 0] // 2 + 1
    cfg = _capi.make_frame_config(gpu_cfg(order), use_fixed=use_fixed)
    got, gres = handle.encode_stereo_frames(x, bps, cfg)
    ocfg = orc.make_frame_config(orc_cfg(order, acorr=orc.ACORR_CANONICAL), use_fixed=use_fixed,
                                 fixed=orc.make_fixed_config(sum_mode=orc.SUMABS_CANONICAL))
    want, wres = orc.encode_stereo_frames_cfg(x, bps, ocfg)
    _check_frames_against_oracle(x, bps, got, gres, want, wres)
    frames = handle.pack_stereo_frames(x, got, gres, bps, 44100, 0, 1)
    for f in range(x.shape[0]):
        assert frames[f] == orc.write_stereo_frame(got[f], x[f, 0], x[f, 1], bps, 44100, f, gres[f, 0], gres[f, 1]), f
    hist = np.bincount(got["kind"].ravel(), minlength=4)
    print("kinds constant/verbatim/fixed/lpc:", hist.tolist(), " assignments:",
          np.bincount(got["channel_assignment"], minlength=4).tolist())
    assert hist[3] > 0 and hist[0] > 0


@pytest.mark.parametrize("channels,bytes_per_sample,total,block", [
    (2, 2, 3 * 4096 + 1234, 4096), (1, 3, 5000, 1152), (8, 2, 4096, 4096), (3, 1, 777, 256), (2, 4, 9000, 4608),
])
def test_fill_le_bytes_equals_reference(handle, channels, bytes_per_sample, total, block):
    """flacenc_hip_fill_le_bytes == le_bytes_to_i32s + deinterleave (arrayutils.rs:273-290, 248-264) per
    FrameBuf (source.rs:288-298), incl. the zero-filled short last block."""
    rng = np.random.default_rng(total)
    data = rng.integers(0, 256, total * channels * bytes_per_sample, dtype=np.uint8).tobytes()
    got = handle.fill_le_bytes(data, channels, bytes_per_sample, block)
    ints = orc.le_bytes_to_i32s(data, bytes_per_sample)
    nf = (total + block - 1) // block
    assert got.shape == (nf, channels, block)
    for f in range(nf):
        chunk = ints[f * block * channels:(f + 1) * block * channels]
        assert np.array_equal(got[f].reshape(-1), orc.deinterleave(chunk, channels, block)), f


# 1596 ..: seeds on which a sweep of 40 000 configurations (tools/fuzz_more.py) once found the RICE2
# flag (bitrepr.rs:540-543) taken from lanes that lead no partition -- 20/24-bit material with
# parameters around 14/15
@pytest.mark.parametrize("seed", list(range(6)) + [1596, 3312, 4204, 4558, 4664, 5313])
def test_frame_pipeline_config_fuzz(handle, seed):
    """Random configurations (order, precision, window, Rice limit, candidate and stereo switches,
    fixed-LPC selector settings, bits per sample, block size) x random material through
    encode_stereo_frames + pack_stereo_frames, everything against the oracle.  Seeded, so a failure
    names its configuration."""
    rng = np.random.default_rng(9000 + seed)
    for trial in range(5):
        n = int(rng.choice([4096, 4096, 4096, 1152, 4608, 256, 2048, 8192, 16384]))  # (8192 / 16384: the big-block frame pipeline, round 4)
        bps = int(rng.choice([8, 12, 16, 16, 20, 24]))
        order = int(rng.choice([1, 2, 4, 6, 8, 8, 10, 12, 12, 16, 24]))
        qcfg = dict(lpc_order=order, quant_precision=int(rng.integers(2, 16)),
                    window=("rectangle" if rng.random() < 0.2 else ("tukey", float(np.round(rng.random(), 2)))),
                    max_rice_parameter=int(rng.choice([0, 3, 7, 14, 15, 30, 30])))
        flags = dict(use_constant=bool(rng.random() < 0.85), use_fixed=bool(rng.random() < 0.7),
                     use_lpc=bool(rng.random() < 0.85), use_leftside=bool(rng.random() < 0.8),
                     use_rightside=bool(rng.random() < 0.8), use_midside=bool(rng.random() < 0.8))
        fixed = dict(fixed_max_order=int(rng.integers(0, 5)), fixed_order_sel=int(rng.random() < 0.75),
                     fixed_partitions=int(rng.choice([1, 2, 4, 8, 16, 16, 32, 64, 3, 12])))
        tag = (seed, trial, n, bps, qcfg, flags, fixed)
        parts = []
        for k in range(3):
            amp = float(rng.choice([0.0, 0.002, 0.1, 0.5, 0.9]))
            namp = float(min(0.99 - amp, rng.choice([0.0, 0.001, 0.05, 0.5])))
            parts.append(_capi.sigen_frames(4, 2, n, bps, float(rng.uniform(2.2, 500.0)), amp, namp,
                                            seed=int(rng.integers(1, 1 << 30))))
        x = np.concatenate(parts)
        x[1, 1] = x[1, 0]
        x[5, 0] = -x[5, 1]
        x[9] = x[9] // 256
        # FLACENC_FUZZ_ORDER=reference|nightly (tools/fuzz_more.py): the same sweep in the reference's summation orders
        mode = os.environ.get("FLACENC_FUZZ_ORDER", "canonical")
        gflag, oac, osum = {"canonical": (0, orc.ACORR_CANONICAL, orc.SUMABS_CANONICAL),
                            "reference": (_capi.FLAG_REFERENCE_SUM_ORDER, orc.ACORR_REFERENCE, orc.SUMABS_STABLE),
                            "nightly": (_capi.FLAG_NIGHTLY_SUM_ORDER, orc.ACORR_NIGHTLY, orc.SUMABS_NIGHTLY)}[mode]
        if mode == "nightly" and order > 15:
            qcfg["lpc_order"] = order = 12
        cfg = _capi.make_frame_config(_capi.make_config(flags=gflag, **qcfg), **flags, **fixed)
        got, gres = handle.encode_stereo_frames(x, bps, cfg)
        ocfg = orc.make_frame_config(orc.make_config(acorr=oac, **qcfg), **flags,
                                     fixed=orc.make_fixed_config(max_order=fixed["fixed_max_order"],
                                                                 order_sel=fixed["fixed_order_sel"],
                                                                 partitions=fixed["fixed_partitions"],
                                                                 sum_mode=osum))
        want, wres = orc.encode_stereo_frames_cfg(x, bps, ocfg)
        try:
            _check_frames_against_oracle(x, bps, got, gres, want, wres)
            frames = handle.pack_stereo_frames(x, got, gres, bps, 48000, 1 << 20, 1)
            for f in range(x.shape[0]):
                assert frames[f] == orc.write_stereo_frame(got[f], x[f, 0], x[f, 1], bps, 48000, (1 << 20) + f,
                                                           gres[f, 0], gres[f, 1]), f
        except AssertionError as e:
            raise AssertionError(f"configuration {tag}: {e}") from e


@pytest.mark.parametrize("seed", range(4))
def test_candidate_and_channel_api_fuzz(handle, seed):
    """Random shapes and configurations through the candidate-level batches (qlpc_batch with
    per-subframe bps, fixed_lpc_batch) and the independent-channel frame calls, against the oracle."""
    rng = np.random.default_rng(7000 + seed)
    for trial in range(4):
        n = int(rng.choice([4096, 4096, 64, 100, 192, 576, 1152, 2304, 4608, 8192, 16384, 20000, 4097, 1000]))
        bps = int(rng.choice([8, 12, 16, 16, 20, 24]))
        order = int(rng.choice([1, 3, 8, 10, 12, 16, 24, 32]))
        order = min(order, 32)
        qcfg = dict(lpc_order=order, quant_precision=int(rng.integers(2, 16)),
                    window=("rectangle" if rng.random() < 0.2 else ("tukey", float(np.round(rng.random(), 2)))),
                    max_rice_parameter=int(rng.choice([0, 5, 14, 15, 30, 30])))
        channels = int(rng.choice([1, 2, 3, 5, 8]))
        nf = int(rng.integers(1, 4))
        amp = float(rng.choice([0.0, 0.003, 0.2, 0.8]))
        namp = float(min(0.99 - amp, rng.choice([0.0, 0.002, 0.1, 0.6])))
        x = _capi.sigen_frames(nf, channels, n, bps, float(rng.uniform(2.2, 400.0)), amp, namp,
                               seed=int(rng.integers(1, 1 << 30)))
        if rng.random() < 0.3:
            x[0, 0] = int(rng.integers(-100, 100))
        tag = (seed, trial, n, bps, channels, nf, qcfg)
        try:
            # --- candidate level: every channel as an independent subframe, mixed bps ---
            flat = x.reshape(-1, n)
            bpsv = np.full(flat.shape[0], bps, np.uint8)
            if bps < 24:
                bpsv[::3] = bps + 1
            # FLACENC_FUZZ_ORDER=reference|nightly (tools/fuzz_more.py): the same sweep in the reference's summation orders
            mode = os.environ.get("FLACENC_FUZZ_ORDER", "canonical")
            gflag, oac, osum = {"canonical": (0, orc.ACORR_CANONICAL, orc.SUMABS_CANONICAL),
                                "reference": (_capi.FLAG_REFERENCE_SUM_ORDER, orc.ACORR_REFERENCE, orc.SUMABS_STABLE),
                                "nightly": (_capi.FLAG_NIGHTLY_SUM_ORDER, orc.ACORR_NIGHTLY, orc.SUMABS_NIGHTLY)}[mode]
            if mode == "nightly" and qcfg["lpc_order"] > 15:
                qcfg["lpc_order"] = 15
            qcfg["flags"] = gflag
            ocfg_kw = {k: v for k, v in qcfg.items() if k != "flags"}
            params, resid, _, _ = handle.qlpc_batch(flat, bpsv, _capi.make_config(**qcfg))
            ocfg = orc.make_config(acorr=oac, **ocfg_kw)
            for k in range(flat.shape[0]):
                w = orc.estimated_qlpc(flat[k], int(bpsv[k]), ocfg)
                p = params[k]
                assert int(p["status"]) == w["status"], k
                if w["status"] == 0:
                    assert int(p["order"]) == w["order"] and int(p["shift"]) == w["shift"], k
                    assert p["coefs"][: w["order"]].tolist() == w["coefs"].tolist(), k
                    assert int(p["rice_order"]) == w["rice_order"] and int(p["code_bits"]) == w["code_bits"], k
                    assert int(p["subframe_bits"]) == w["subframe_bits"] and int(p["sum_quotients"]) == w["sum_quotients"], k
                    assert np.array_equal(resid[k], w["residual"]), k
            fx = dict(fixed_max_order=int(rng.integers(0, 5)), fixed_order_sel=int(rng.random() < 0.7),
                      fixed_partitions=int(rng.integers(1, 65)))
            fcfg = _capi.make_frame_config(_capi.make_config(**qcfg), use_fixed=True, **fx)
            fp, fr, fk = handle.fixed_lpc_batch(flat, bpsv, fcfg)
            ofx = orc.make_fixed_config(max_order=fx["fixed_max_order"], order_sel=fx["fixed_order_sel"],
                                        partitions=fx["fixed_partitions"], sum_mode=osum)
            for k in range(flat.shape[0]):
                w = orc.fixed_lpc(flat[k], int(bpsv[k]), 2 ** 63, ofx, max_p=int(ocfg.max_rice_parameter))
                assert int(fp[k]["order"]) == w["order"] and int(fk[k]) == w["estimate"][w["order"]], (k, fx)
                assert int(fp[k]["subframe_bits"]) == w["subframe_bits"] and np.array_equal(fr[k], w["residual"]), (k, fx)
            # --- frame level, independent channels ---
            flags = dict(use_constant=bool(rng.random() < 0.8), use_lpc=bool(rng.random() < 0.85))
            fcfg = _capi.make_frame_config(_capi.make_config(**qcfg), use_fixed=bool(rng.random() < 0.7), **flags, **fx)
            res, rr = handle.encode_frames(x, bps, fcfg)
            ofc = orc.make_frame_config(ocfg, use_fixed=bool(fcfg.use_fixed), fixed=ofx, **flags)
            if channels * (8 + n * bps) // 8 + 64 > 150 * 1024:
                # the frame would not fit the packer's LDS bit buffer: a documented UNSUPPORTED
                with pytest.raises(_capi.FlacencHipError) as ei:
                    handle.pack_frames(x, res, rr, bps, 44100, 3, 2)
                assert ei.value.code == _capi.ERR_UNSUPPORTED
                packed = None
            else:
                packed = handle.pack_frames(x, res, rr, bps, 44100, 3, 2)
            for f in range(nf):
                subs = []
                for c in range(channels):
                    w = orc.encode_subframe(x[f, c], bps, ofc)
                    g = res[f, c]
                    assert int(g["kind"]) == w["kind"] and int(g["bits"]) == w["bits"], (f, c, fx, flags)
                    if w["kind"] >= 2:
                        assert np.array_equal(rr[f, c], w["residual"]), (f, c)
                    pz = g["params"]
                    subs.append(dict(kind=int(g["kind"]), bps=bps, samples=x[f, c], dc_offset=int(g["dc_offset"]),
                                     order=int(pz["order"]), shift=int(pz["shift"]), precision=int(pz["precision"]),
                                     coefs=pz["coefs"], rice_order=int(pz["rice_order"]), rice_params=pz["rice_params"],
                                     residual=rr[f, c]))
                assert packed is None or packed[f] == orc.write_frame(n, 0, bps, 44100, 3 + 2 * f, subs), f
        except AssertionError as e:
            raise AssertionError(f"configuration {tag}: {e}") from e


def _extreme_frames(rng, n, bps):
    """Stereo frames built from worst-case material: full-scale alternation and square waves,
    impulses, full-range ramps, clipped sines, one channel silent / constant / inverted."""
    lo, hi = -(1 << (bps - 1)), (1 << (bps - 1)) - 1
    t = np.arange(n)

    def one():
        k = int(rng.integers(0, 8))
        if k == 0:
            return np.where(t % 2 == 0, hi, lo)
        if k == 1:
            return np.where((t // int(rng.integers(1, 200))) % 2 == 0, hi, lo)
        if k == 2:
            x = np.zeros(n, np.int64)
            x[rng.integers(0, n, int(rng.integers(1, 6)))] = rng.choice([lo, hi])
            return x
        if k == 3:
            return np.linspace(lo, hi, n).astype(np.int64)
        if k == 4:
            return np.clip(np.sin(t / float(rng.uniform(1.5, 300.0))) * hi * float(rng.uniform(1.0, 4.0)), lo, hi).astype(np.int64)
        if k == 5:
            return np.full(n, int(rng.integers(lo, hi + 1)))
        if k == 6:
            return rng.integers(lo, hi + 1, n)
        return (rng.integers(-3, 4, n)).cumsum().clip(lo, hi)

    frames = []
    for _ in range(6):
        l, r = one(), one()
        m = int(rng.integers(0, 5))
        if m == 0:
            r = l.copy()
        elif m == 1:
            r = np.clip(-l, lo, hi)
        elif m == 2:
            r = np.clip(l + rng.integers(-2, 3, n), lo, hi)
        frames.append(np.stack([l, r]))
    return np.stack(frames).astype(np.int32)


# 2616: full-scale 24-bit alternation whose order-24 residual wraps i32 -- the reference's u32-wrapping
# table sums (rice.rs:88-93) stay small while the true quotient sum is ~2e12 bits; found by the sweep
@pytest.mark.parametrize("seed", list(range(4)) + [2616])
def test_extreme_signals_and_layout_fuzz(handle, monkeypatch, seed):
    """Worst-case material (the i64 residual path, saturating Rice tables, RICE2 parameters, constant
    and verbatim subframes) through the device-pointer entry points with random row strides and
    misaligned base pointers (which must push block-4096 work onto the general path), against the
    oracle."""
    import torch
    rng = np.random.default_rng(5000 + seed)
    for trial in range(4):
        n = int(rng.choice([4096, 4096, 4096, 1152, 4608, 512]))
        bps = int(rng.choice([8, 16, 16, 24]))
        order = int(rng.choice([1, 4, 8, 12, 24]))
        qcfg = dict(lpc_order=order, quant_precision=int(rng.integers(3, 16)),
                    window=("rectangle" if rng.random() < 0.3 else ("tukey", float(np.round(rng.random(), 2)))),
                    max_rice_parameter=int(rng.choice([0, 4, 14, 15, 30, 30])))
        use_fixed = bool(rng.random() < 0.6)
        fx = dict(fixed_max_order=int(rng.integers(0, 5)), fixed_order_sel=int(rng.random() < 0.7),
                  fixed_partitions=int(rng.choice([1, 4, 16, 64, 7])))
        x = _extreme_frames(rng, n, bps)
        F = x.shape[0]
        stride = n + int(rng.choice([0, 0, 4, 8, 3, 5]))
        rstride = n + int(rng.choice([0, 0, 4, 1]))
        off_in, off_out = int(rng.choice([0, 0, 4, 1, 2])), int(rng.choice([0, 0, 4, 3]))
        tag = (seed, trial, n, bps, qcfg, use_fixed, fx, stride, rstride, off_in, off_out)
        buf = torch.zeros(F * 2 * stride + 8, dtype=torch.int32, device="cuda")
        view = buf[off_in:off_in + F * 2 * stride].view(F * 2, stride)
        view[:, :n] = torch.from_numpy(x.reshape(F * 2, n)).cuda()
        res = torch.zeros((F, 752), dtype=torch.uint8, device="cuda")
        rbuf = torch.full((F * 2 * rstride + 8,), -7, dtype=torch.int32, device="cuda")
        rview = rbuf[off_out:off_out + F * 2 * rstride].view(F * 2, rstride)
        cfg = _capi.make_frame_config(_capi.make_config(flags=_capi.FLAG_FUSED_PACK, **qcfg), use_fixed=use_fixed, **fx)
        handle.encode_stereo_frames_device(cfg, view.data_ptr(), F, n, stride, bps, res.data_ptr(), rview.data_ptr(),
                                           rstride, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        got = np.frombuffer(res.cpu().numpy().tobytes(), dtype=_capi.FRAME_RESULT_DTYPE)
        gres = rview[:, :n].cpu().numpy().reshape(F, 2, n)
        ocfg = orc.make_frame_config(orc.make_config(acorr=orc.ACORR_CANONICAL, **qcfg), use_fixed=use_fixed,
                                     fixed=orc.make_fixed_config(max_order=fx["fixed_max_order"], order_sel=fx["fixed_order_sel"],
                                                                 partitions=fx["fixed_partitions"], sum_mode=orc.SUMABS_CANONICAL))
        want, wres = orc.encode_stereo_frames_cfg(x, bps, ocfg)
        # the one-call PCM -> frame bytes path (the fused bit writer where the shape allows it)
        ostride = handle.frame_bytes_bound(n, bps)
        out = torch.zeros((F, ostride), dtype=torch.uint8, device="cuda")
        lens = torch.zeros(F, dtype=torch.int32, device="cuda")
        res2 = torch.zeros((F, 752), dtype=torch.uint8, device="cuda")
        first = int(rng.choice([0, 127, 128, 1 << 11, 1 << 16, 1 << 21, 1 << 26, (1 << 31) - F]))
        handle.encode_pack_stereo_frames_device(cfg, view.data_ptr(), F, n, stride, bps, 44100, first, 1,
                                                res2.data_ptr(), out.data_ptr(), ostride, lens.data_ptr(),
                                                stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        o, ln = out.cpu().numpy(), lens.cpu().numpy()
        try:
            _check_frames_against_oracle(x, bps, got, gres, want, wres)
            if rstride > n:
                assert bool((rview[:, n:] == -7).all()), "wrote beyond the block"
            _decode_frames(x, got, gres)
            assert res2.cpu().numpy().tobytes() == res.cpu().numpy().tobytes()
            for f in range(F):
                wantb = orc.write_stereo_frame(want[f], x[f, 0], x[f, 1], bps, 44100, first + f, wres[f, 0], wres[f, 1])
                assert bytes(o[f, :ln[f]]) == wantb, ("packed frame", f, int(ln[f]), len(wantb))
        except AssertionError as e:
            raise AssertionError(f"configuration {tag}: {e}") from e


@pytest.mark.parametrize("n,bps,order,use_fixed,first,step", [
    (4096, 16, 8, True, 0, 1), (4096, 16, 12, False, 1 << 21, 3), (4096, 24, 10, True, 100, 1),
    (4096, 8, 8, True, 5, 1), (1152, 16, 8, True, 0, 1), (4096, 16, 16, True, 7, 1),
])
def test_fused_encode_and_pack_equals_two_stage_path(handle, monkeypatch, n, bps, order, use_fixed, first, step):
    """flacenc_hip_encode_pack_stereo_frames_async (for 4096-sample blocks one kernel: the residual
    never reaches HBM) == encode_stereo_frames followed by pack_stereo_frames == the oracle's
    controller and bit writer, for every subframe kind and channel assignment."""
    import torch
    x = np.ascontiguousarray(_fixed_corpus()[:, :, :n])
    if bps == 24:
        x = (x.astype(np.int64) * 181).astype(np.int32)
        x[4] = np.stack([util.quantize(util.noise(21, n, 0.999), 24), util.quantize(util.noise(22, n, 0.999), 24)])
    if bps == 8:
        x = (x // 256).astype(np.int32)
    F = x.shape[0]
    # one kernel also without the fixed-LPC candidate (the default is two-stage there)
    cfg = _capi.make_frame_config(_capi.make_config(lpc_order=order, flags=_capi.FLAG_FUSED_PACK), use_fixed=use_fixed)
    xs = torch.from_numpy(x).cuda()
    res = torch.zeros((F, 752), dtype=torch.uint8, device="cuda")
    stride = handle.frame_bytes_bound(n, bps)
    out = torch.zeros((F, stride), dtype=torch.uint8, device="cuda")
    lens = torch.zeros(F, dtype=torch.int32, device="cuda")
    handle.encode_pack_stereo_frames_device(cfg, xs.data_ptr(), F, n, n, bps, 44100, first, step, res.data_ptr(),
                                            out.data_ptr(), stride, lens.data_ptr(),
                                            stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = np.frombuffer(res.cpu().numpy().tobytes(), dtype=_capi.FRAME_RESULT_DTYPE)
    o, ln = out.cpu().numpy(), lens.cpu().numpy()
    want, wres = handle.encode_stereo_frames(x, bps, cfg)
    frames = handle.pack_stereo_frames(x, want, wres, bps, 44100, first, step)
    kinds = set()
    for f in range(F):
        assert got[f].tobytes() == want[f].tobytes(), f
        assert bytes(o[f, :ln[f]]) == frames[f], (f, int(ln[f]), len(frames[f]))
        assert frames[f] == orc.write_stereo_frame(want[f], x[f, 0], x[f, 1], bps, 44100, first + f * step,
                                                   wres[f, 0], wres[f, 1]), f
        kinds.update(want[f]["kind"].tolist())
    if n == 4096 and bps != 8:
        assert kinds >= ({0, 1, 2, 3} if use_fixed else {0, 1, 3})


@pytest.mark.parametrize("seed", range(4))
def test_extreme_candidates_fuzz(handle, seed):
    """Worst-case material through the candidate-level batches for random block sizes and orders up
    to 32: QLPC records, fixed-LPC records and the independent-channel frame decisions vs the oracle."""
    rng = np.random.default_rng(3000 + seed)
    for trial in range(4):
        n = int(rng.choice([4096, 4096, 64, 97, 192, 1152, 4608, 8192, 16384, 4000, 17000]))
        bps = int(rng.choice([8, 16, 16, 24, 24]))
        order = int(rng.choice([1, 2, 8, 12, 16, 24, 32]))
        qcfg = dict(lpc_order=order, quant_precision=int(rng.integers(2, 16)),
                    window=("rectangle" if rng.random() < 0.3 else ("tukey", float(np.round(rng.random(), 2)))),
                    max_rice_parameter=int(rng.choice([0, 4, 14, 15, 30, 30])))
        x = _extreme_frames(rng, n, bps)          # [6, 2, n]
        flat = x.reshape(-1, n)
        bpsv = np.full(flat.shape[0], bps, np.uint8)
        if bps < 24:
            bpsv[1::2] = bps + 1
        tag = (seed, trial, n, bps, qcfg)
        try:
            params, resid, _, _ = handle.qlpc_batch(flat, bpsv, _capi.make_config(**qcfg))
            ocfg = orc.make_config(acorr=orc.ACORR_CANONICAL, **qcfg)
            for k in range(flat.shape[0]):
                w = orc.estimated_qlpc(flat[k], int(bpsv[k]), ocfg)
                p = params[k]
                assert int(p["status"]) == w["status"], (k, "status")
                if w["status"] == 0:
                    for fld in ("order", "shift", "rice_order", "code_bits", "subframe_bits", "sum_quotients"):
                        assert int(p[fld]) == int(w[fld]), (k, fld, int(p[fld]), int(w[fld]))
                    assert p["coefs"][: w["order"]].tolist() == w["coefs"].tolist(), (k, "coefs")
                    assert np.array_equal(resid[k], w["residual"]), (k, "residual")
            fx = dict(fixed_max_order=int(rng.integers(0, 5)), fixed_order_sel=int(rng.random() < 0.7),
                      fixed_partitions=int(rng.integers(1, 65)))
            fcfg = _capi.make_frame_config(_capi.make_config(**qcfg), use_fixed=True, **fx)
            fp, fr, fk = handle.fixed_lpc_batch(flat, bpsv, fcfg)
            ofx = orc.make_fixed_config(max_order=fx["fixed_max_order"], order_sel=fx["fixed_order_sel"],
                                        partitions=fx["fixed_partitions"], sum_mode=orc.SUMABS_CANONICAL)
            for k in range(flat.shape[0]):
                w = orc.fixed_lpc(flat[k], int(bpsv[k]), 2 ** 63, ofx, max_p=int(ocfg.max_rice_parameter))
                assert int(fp[k]["order"]) == w["order"] and int(fk[k]) == w["estimate"][w["order"]], (k, fx, "fixed order/key")
                assert int(fp[k]["subframe_bits"]) == w["subframe_bits"], (k, fx, "fixed bits", int(fp[k]["subframe_bits"]), w["subframe_bits"])
                assert np.array_equal(fr[k], w["residual"]), (k, fx, "fixed residual")
            res, rr = handle.encode_frames(x, bps, fcfg)
            ofc = orc.make_frame_config(ocfg, use_fixed=True, fixed=ofx)
            for f in range(x.shape[0]):
                for c in range(2):
                    w = orc.encode_subframe(x[f, c], bps, ofc)
                    assert int(res[f, c]["kind"]) == w["kind"] and int(res[f, c]["bits"]) == w["bits"], \
                        (f, c, fx, int(res[f, c]["kind"]), w["kind"], int(res[f, c]["bits"]), w["bits"])
        except AssertionError as e:
            raise AssertionError(f"configuration {tag}: {e}") from e


@pytest.mark.parametrize("channels,n", [(1, 4096), (6, 4096), (3, 1000)])
def test_one_call_independent_channel_pipeline(handle, channels, n):
    """flacenc_hip_encode_pack_frames_async == encode_frames + pack_frames."""
    import torch
    bps = 16
    x = _capi.sigen_frames(5, channels, n, bps, 77.0, 0.4, 0.03, seed=channels + n)
    x[2, 0] = (np.arange(n) // 5) % 3000
    cfg = _capi.make_frame_config(gpu_cfg(8), use_fixed=True)
    want, wres = handle.encode_frames(x, bps, cfg)
    frames = handle.pack_frames(x, want, wres, bps, 32000, 9, 1)
    xs = torch.from_numpy(x).cuda()
    res = torch.zeros((5 * channels, 368), dtype=torch.uint8, device="cuda")
    stride = int(handle._lib.flacenc_hip_frame_bytes_bound(channels, n, bps))
    out = torch.zeros((5, stride), dtype=torch.uint8, device="cuda")
    lens = torch.zeros(5, dtype=torch.int32, device="cuda")
    handle.encode_pack_frames_device(cfg, xs.data_ptr(), 5, channels, n, n, bps, 32000, 9, 1, res.data_ptr(),
                                     out.data_ptr(), stride, lens.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert res.cpu().numpy().tobytes() == want.tobytes()
    o, ln = out.cpu().numpy(), lens.cpu().numpy()
    for f in range(5):
        assert bytes(o[f, :ln[f]]) == frames[f], f


@pytest.mark.parametrize("n,bps,order,use_fixed", [(4096, 16, 8, True), (4096, 16, 12, False), (4096, 24, 10, True),
                                                   (4608, 16, 8, True), (8192, 24, 24, False), (1152, 16, 10, True)])
def test_finest_rice_order_extension(handle, n, bps, order, use_fixed):
    """FLACENC_HIP_FLAG_FINEST_RICE_ORDER (a build extension, BASELINE config 2's "fixed Rice partition
    order"): the search keeps the finest partition order.  Same restriction in the oracle -> identical
    decisions, records, residuals and packed bytes; the frames are valid FLAC (independent parser); and no
    frame is smaller than what the exhaustive search gives."""
    import flac_parse
    x = np.ascontiguousarray(_fixed_corpus()[:12, :, :n]) if n <= 4096 else _capi.sigen_frames(6, 2, n, 16, 90.0, 0.4, 0.05, seed=n)
    if bps == 24:
        x = (x.astype(np.int64) * 181).astype(np.int32)
    cfg = _capi.make_frame_config(_capi.make_config(lpc_order=order, rice_finest_only=True), use_fixed=use_fixed)
    got, gres = handle.encode_stereo_frames(x, bps, cfg)
    ocfg = orc.make_frame_config(orc.make_config(lpc_order=order, acorr=orc.ACORR_CANONICAL, rice_finest_only=True),
                                 use_fixed=use_fixed, fixed=orc.make_fixed_config(sum_mode=orc.SUMABS_CANONICAL))
    want, wres = orc.encode_stereo_frames_cfg(x, bps, ocfg)
    _check_frames_against_oracle(x, bps, got, gres, want, wres)
    full, _ = handle.encode_stereo_frames(x, bps, _capi.make_frame_config(_capi.make_config(lpc_order=order), use_fixed=use_fixed))
    frames = handle.pack_stereo_frames(x, got, gres, bps, 44100, 0, 1)
    fo = {4096: 6, 4608: 6, 8192: 7, 1152: 4}[n]
    for f in range(x.shape[0]):
        for c in range(2):
            if int(got[f]["kind"][c]) >= 2:
                assert int(got[f]["lpc"][c]["rice_order"]) == fo
        assert sum(int(got[f]["bits"][r]) for r in got[f]["role"]) >= sum(int(full[f]["bits"][r]) for r in full[f]["role"])
        assert frames[f] == orc.write_stereo_frame(got[f], x[f, 0], x[f, 1], bps, 44100, f, gres[f, 0], gres[f, 1])
        if n <= 4608 and f < 4:
            assert np.array_equal(flac_parse.parse_frame(frames[f], stream_bps=bps)["channels"], x[f])


def test_encode_stereo_frames_rejects_bad_config(handle):
    x = np.zeros((2, 2, 4096), np.int32)
    with pytest.raises(_capi.FlacencHipError) as ei:
        handle.encode_stereo_frames(x, 16, _capi.make_frame_config(gpu_cfg(8), use_fixed=True, fixed_max_order=5))
    assert ei.value.code == _capi.ERR_BAD_CONFIG
    with pytest.raises(_capi.FlacencHipError) as ei:
        handle.encode_stereo_frames(x, 16, _capi.make_frame_config(gpu_cfg(8), use_fixed=True, fixed_partitions=65))
    assert ei.value.code == _capi.ERR_BAD_CONFIG
    # (blocks below 64 samples are taken by the frame-level calls since round 3 -- test_blocks_shorter_than_the_
    # prediction_minimum; the candidate-level batch still refuses them, as does any call above 32767)
    with pytest.raises(_capi.FlacencHipError) as ei:
        handle.stereo_qlpc_batch(np.zeros((2, 2, 32), np.int32), 16, gpu_cfg(8))
    assert ei.value.code == _capi.ERR_BAD_ARGUMENT
    with pytest.raises(_capi.FlacencHipError) as ei:
        handle.encode_stereo_frames(np.zeros((1, 2, 32768), np.int32), 16, _capi.make_frame_config(gpu_cfg(8)))
    assert ei.value.code == _capi.ERR_BAD_ARGUMENT


@pytest.mark.parametrize("n", [1, 2, 15, 16, 33, 63])
@pytest.mark.parametrize("channels", [1, 2, 5])
def test_blocks_shorter_than_the_prediction_minimum(handle, n, channels):
    """Blocks of 1..63 samples (a stream's last block): too_short in encode_subframe (src/coding.rs:389-418) -- no
    fixed_lpc, no estimated_qlpc, Constant or Verbatim -- through the frame-level calls; decisions and frame bytes
    == the oracle's controller and writer, and the bytes parse back to the input."""
    import flac_parse
    bps = 16
    x = _capi.sigen_frames(4, channels, n, bps, 40.0, 0.5, 0.1, seed=100 * n + channels)
    x[1, 0] = -3                      # Constant
    if channels >= 2:
        x[2, 1] = x[2, 0]             # side channel constant 0
    cfg = _capi.make_frame_config(gpu_cfg(8), use_fixed=True)
    ocfg = orc.make_frame_config(orc_cfg(8, acorr=orc.ACORR_CANONICAL), use_fixed=True,
                                 fixed=orc.make_fixed_config(sum_mode=orc.SUMABS_CANONICAL))
    if channels == 2:
        res, resid = handle.encode_stereo_frames(x, bps, cfg)
        want, wres = orc.encode_stereo_frames_cfg(x, bps, ocfg)
        assert res["channel_assignment"].tolist() == want["channel_assignment"].tolist()
        assert res["kind"].tolist() == want["kind"].tolist() and res["bits"].tolist() == want["bits"].tolist()
        assert set(res["kind"].ravel().tolist()) <= {0, 1} and not resid.any()
        packed = handle.pack_stereo_frames(x, res, resid, bps, 44100, 9, 1)
        for f in range(x.shape[0]):
            assert packed[f] == orc.write_stereo_frame(res[f], x[f, 0], x[f, 1], bps, 44100, 9 + f,
                                                       resid[f, 0], resid[f, 1]), f
    else:
        res, resid = handle.encode_frames(x, bps, cfg)
        packed = handle.pack_frames(x, res, resid, bps, 44100, 9, 1)
        for f in range(x.shape[0]):
            subs = []
            for c in range(channels):
                w = orc.encode_subframe(x[f, c], bps, ocfg)
                assert int(res[f, c]["kind"]) == w["kind"] and int(res[f, c]["bits"]) == w["bits"] and w["kind"] <= 1
                subs.append(dict(kind=w["kind"], bps=bps, samples=x[f, c], dc_offset=int(res[f, c]["dc_offset"])))
            assert packed[f] == orc.write_frame(n, 0, bps, 44100, 9 + f, subs), f
    for f in range(x.shape[0]):
        got = flac_parse.parse_frame(packed[f], stream_bps=bps, stream_rate=44100)
        assert got["block_size"] == n and got["number"] == 9 + f and np.array_equal(got["channels"], x[f]), f
