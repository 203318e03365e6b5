"""Pins the CPU oracle against the reference's own in-source known-answer tests.

Every test cites the reference test it restates (paths relative to
/root/reference/).  Inputs and expected values are the reference's test data.
Where the reference test draws from `sigen::Noise` (rand's StdRng, not
reproducible here) the same property is asserted on counter-based noise.
"""
import numpy as np
import pytest

import util
from oracle import oracle as orc


def assert_close(actual, expected, rtol=1e-5, atol=1e-5):
    """assert_close!, src/test_helper.rs:46-56."""
    assert abs(actual - expected) < rtol * abs(expected) + atol, (actual, expected)


# ---------------------------------------------------------------- lpc.rs ----
def test_auto_correlation_computation():
    """src/lpc.rs:997-1022 (T = f32): 128-sample sine of period 32 -> argmax 0, argmin 16."""
    t = np.arange(128, dtype=np.float32)
    signal = (np.sin((t / np.float32(32.0) * np.float32(2.0) * np.float32(np.pi)).astype(np.float32))
              .astype(np.float32) * np.float32(1024.0)).astype(np.float32)
    corr = orc.auto_correlation(32, signal, dtype=np.float32)
    assert int(np.argmax(corr)) == 0
    assert int(np.argmin(corr)) == 16


KNOWN = [0.0] * 8 + [1, 1, 1, 1, -1, -1, -1, -1, 1, 1, -1, -1, 1, 1, -1, -1,
                     1, -1, 1, -1, 1, -1, 1, -1,
                     1, -1, 1, -1, 1, -1, 1, -1, 1, 1, -1, -1, 1, 1, -1, -1,
                     1, 1, 1, 1, -1, -1, -1, -1] + [0.0] * 8


@pytest.mark.parametrize("canonical", [False, True])
def test_auto_correlation_computation_with_known_samples(canonical):
    """src/lpc.rs:1024-1041: 64-sample +-1 pattern, order 33 -> 24, -4, 2, ..., R[32] = 0.
    All partial sums are small integers, so the canonical order must agree exactly too."""
    assert len(KNOWN) == 64
    corr = orc.auto_correlation(33, np.array(KNOWN, np.float32), canonical=canonical)
    assert corr[0] == 24.0
    assert corr[1] == -4.0
    assert corr[2] == 2.0
    assert corr[32] == 0.0


def test_symmetric_levinson_algorithm():
    """src/lpc.rs:1043-1066."""
    xs, st = orc.symmetric_levinson_recursion([1.0, 0.5, 0.0, 0.25], [1.0, -1.0, 1.0, -1.0],
                                              dtype=np.float32)
    assert st == 0
    assert xs.tolist() == [8.0, -10.0, 10.0, -8.0]  # exact in f32
    xs, st = orc.symmetric_levinson_recursion([1.0, -0.5, -1.0, -0.5, 0.5],
                                              [1.0, 0.5, 0.25, 0.125, 0.0625], dtype=np.float32)
    for x, e in zip(xs, [0.80833, -0.26458, -0.36667, -0.45208, -1.06667]):
        assert_close(float(x), e)
    # the f64 instantiation the path uses (lpc.rs:916) must agree to the same tolerance
    xs64, _ = orc.symmetric_levinson_recursion([1.0, -0.5, -1.0, -0.5, 0.5],
                                               [1.0, 0.5, 0.25, 0.125, 0.0625])
    for x, e in zip(xs64, [0.80833, -0.26458, -0.36667, -0.45208, -1.06667]):
        assert_close(float(x), e)


def test_shift_finder():
    """src/lpc.rs:1068-1074."""
    assert orc.find_shift([0.25, 0.125, 0.000001, 0.0], 8) == 9


def test_parameter_quantizer():
    """src/lpc.rs:1076-1086."""
    qp = orc.quantize_parameters([0.0, 0.5, 0.1], 4)
    assert list(qp.coefs[: qp.order]) == [0, 7, 2]
    qp = orc.quantize_parameters([1.0, -0.5, 0.5], 2)
    assert list(qp.coefs[: qp.order]) == [1, -1, 1]
    # QuantizedParameters::dequantized, src/component/datatype.rs:2174-2177, 2258-2263
    deq = [np.float32(c) * np.float32(2.0) ** np.float32(-qp.shift) for c in qp.coefs[: qp.order]]
    assert deq == [0.5, -0.5, 0.5]


def test_qlpc_auto_truncation():
    """src/lpc.rs:1088-1093."""
    assert orc.quantize_parameters([1.0, 0.5, 0.0, 0.0], 8).order == 2


@pytest.mark.parametrize("lpc_order", [2, 12, 24])
def test_qlpc_recovery(lpc_order):
    """src/lpc.rs:1095-1143: Sine(32, 0.8) + noise(0.01), 16-bit, 1024 samples, Tukey(0.1),
    precision 15: error energy < signal energy and e[t] + (sum c*s >> shift) == s[t]."""
    signal = util.sine_noise(1024, 16, 32, 0.8, 0.01, seed=123)
    cfg = orc.make_config(lpc_order=lpc_order, quant_precision=15, window=("tukey", 0.1))
    _, coefs, st = orc.lpc_from_autocorr(signal, cfg)
    assert st == 0 and np.isfinite(coefs).all()
    qlpc = orc.quantize_parameters(coefs, 15)
    assert qlpc.order <= lpc_order
    errors = orc.compute_error(qlpc, signal)
    sig_e = float((signal[lpc_order:].astype(np.float64) ** 2).sum())
    err_e = float((errors[lpc_order:].astype(np.float64) ** 2).sum())
    assert err_e < sig_e
    qc = [int(c) for c in qlpc.coefs[: qlpc.order]]
    for t in range(lpc_order, len(signal)):
        pred = sum(int(signal[t - tau - 1]) * c for tau, c in enumerate(qc)) >> qlpc.shift
        assert int(errors[t]) + pred == int(signal[t]), t


def test_lpc_with_pure_dc():
    """src/lpc.rs:1145-1169 (f32 estimator, order 1)."""
    signal = np.array([12345] * 7, np.int32)
    corr = orc.auto_correlation(2, signal.astype(np.float32), dtype=np.float32)
    coefs, _ = orc.symmetric_levinson_recursion(corr[:1], corr[1:2], dtype=np.float32)
    assert_close(float(coefs[0]), 1.0)
    qlpc = orc.quantize_parameters(coefs.astype(np.float64), 15)
    errors = orc.compute_error(qlpc, signal)
    assert (errors < 2).all()


def test_lpc_with_known_coefs():
    """src/lpc.rs:1171-1192: sign pattern (+, -, +) of the order-3 estimate."""
    signal = [0, -512, 0, 512, 256, -256, -256, 128, 256, 0, -192, -64, 128, 96, -64, -96, 16,
              80, 16, -56, -32, 32, 36, -12]
    cfg = orc.make_config(lpc_order=3, window=("tukey", 0.25))
    _, coefs, st = orc.lpc_from_autocorr(np.array(signal, np.int32), cfg)
    assert st == 0
    assert coefs[0] > 0.0 and coefs[1] < 0.0 and coefs[2] > 0.0


def test_tukey_window():
    """src/lpc.rs:1214-1228: Tukey(0.3, 32) against scipy.signal.windows.tukey(32, 0.3)."""
    reference = [0., 0.1098376, 0.39109322, 0.720197, 0.95255725] + [1.] * 22 + \
                [0.95255725, 0.720197, 0.39109322, 0.1098376, 0.]
    win = orc.window_weights(("tukey", 0.3), 32)
    for w, e in zip(win, reference):
        assert_close(float(w), e)


def test_tukey_window_range():
    """src/lpc.rs:1230-1243: every weight is normal or zero for alpha in {0, .3, .5, .8, 1}."""
    tiny = np.finfo(np.float32).tiny
    for alpha in [0.0, 0.3, 0.5, 0.8, 1.0]:
        w = orc.window_weights(("tukey", alpha), 4096)
        assert np.isfinite(w).all()
        assert ((np.abs(w) >= tiny) | (w == 0.0)).all()
    assert (orc.window_weights(("tukey", 0.0), 64) == 1.0).all()  # lpc.rs:99-101
    assert (orc.window_weights("rectangle", 64) == 1.0).all()
    assert orc.window_weights(("tukey", 0.4), 4096)[0] == 0.0  # t == 0 -> exactly 0


def test_qlpc_with_test_signal():
    """src/lpc.rs:1258-1295: sus109 ch0, 4096 samples, order 8, precision 12, Tukey(0.1)."""
    signal = util.test_signal("sus109", 0)[:4096]
    cfg = orc.make_config(lpc_order=8, quant_precision=12, window=("tukey", 0.1))
    _, coefs, st = orc.lpc_from_autocorr(signal, cfg)
    assert st == 0
    qlpc = orc.quantize_parameters(coefs, 12)
    assert qlpc.order == 8
    errors = orc.compute_error(qlpc, signal)
    sig_e = float((signal[8:].astype(np.float64) ** 2).sum())
    err_e = float((errors[8:].astype(np.float64) ** 2).sum())
    assert err_e < sig_e


def test_overflow_patterns():
    """src/lpc.rs:1415-1429: must not crash on the i64 fallback; plus losslessness."""
    signal = np.array([127] * 33 + [29] + [0] * 30, np.int32)
    cfg = orc.make_config(lpc_order=15, quant_precision=13, window="rectangle")
    _, coefs, st = orc.lpc_from_autocorr(signal, cfg)
    assert st == 0
    qlpc = orc.quantize_parameters(coefs, 13)
    errors = orc.compute_error(qlpc, signal)
    dec = orc.decode_lpc(signal[: qlpc.order], qlpc.coefs[: qlpc.order], qlpc.shift, errors)
    assert (dec == signal).all()


def test_order_zero_lpc():
    """src/lpc.rs:1431-1446."""
    signal = np.zeros(64, np.int32)
    cfg = orc.make_config(lpc_order=0, quant_precision=13, window="rectangle")
    _, coefs, st = orc.lpc_from_autocorr(signal, cfg)
    assert st == 0 and len(coefs) == 0
    qlpc = orc.quantize_parameters([], 13)
    assert qlpc.order == 0
    assert (orc.compute_error(qlpc, signal) == 0).all()


def test_levinson_zero_denominator_skips_iteration():
    """src/lpc.rs:678-682: `continue` targets the `for`, so iteration n is skipped.
    R = [1, 1, ...] gives error = 1, denom = fma(1, -1, 1) = 0 at n = 1."""
    xs, st = orc.symmetric_levinson_recursion([1.0, 1.0, 1.0], [1.0, 1.0, 1.0])
    assert st == 0
    assert np.isfinite(xs).all()
    assert xs[0] == 1.0  # dest[0] = ys[0] / coefs[0]; iterations 1 and 2 both skipped
    assert xs[1] == 0.0 and xs[2] == 0.0


def test_compute_error_i32_and_i64_paths_agree():
    """src/lpc.rs:373-389: both branches produce the exact value truncated to i32."""
    sig16 = util.sine_noise(512, 16, 50, 0.3, 0.01, seed=5)
    sig24 = util.sine_noise(512, 24, 50, 0.9, 0.05, seed=6)
    qp = orc.qparams([12000, -9000, 4000, -1500], 13, 15)
    for sig in (sig16, sig24):
        e = orc.compute_error(qp, sig)
        for t in range(4, len(sig)):
            pred = sum(int(sig[t - 1 - j]) * int(qp.coefs[j]) for j in range(4)) >> 13
            want = (int(sig[t]) - pred + (1 << 31)) % (1 << 32) - (1 << 31)
            assert int(e[t]) == want
        assert (e[:4] == 0).all()


# --------------------------------------------------------------- rice.rs ----
def test_bit_table_initialization():
    """src/rice.rs:319-324."""
    table = orc.prc_bit_table_from_errors([6, 8, 10, 12], 4)
    assert table[0] == 3 * 2 + 4 * 2 + 5 * 2 + 6 * 2 + 8
    assert table[1] == 3 + 4 + 5 + 6 + 8 + 4


def test_prc_parameter_search():
    """src/rice.rs:326-339 (property on 12-bit noise of amplitude 0.25, 64 samples)."""
    signal = util.quantize(util.noise(11, 64, 0.25), 12)
    errors = [orc.encode_signbit(int(v)) for v in signal]
    p, _ = orc.prc_minimizer(orc.prc_bit_table_from_errors(errors, 4), 14)
    assert 0 < p < 14


def test_finest_partition_order_search():
    """src/rice.rs:341-349."""
    assert orc.finest_partition_order(64, 4) == 4
    assert orc.finest_partition_order(64, 3) == 4
    assert orc.finest_partition_order(192, 1) == 6
    assert orc.finest_partition_order(192, 3) == 6
    assert orc.finest_partition_order(192, 4) == 5
    # sizes the path sees
    assert [orc.finest_partition_order(n, 64) for n in (4096, 8192, 16384)] == [6, 7, 8]


def test_partitioned_rice_parameter_search():
    """src/rice.rs:351-365: loud half + quiet half, 8-bit, 128 samples, warm-up 4."""
    signal = util.quantize(np.concatenate([util.noise(0, 64, 0.5), util.noise(1, 64, 0.05)]), 8)
    errors = np.array([orc.encode_signbit(int(v)) for v in signal], np.uint32)
    _, single_bits = orc.prc_minimizer(orc.prc_bit_table_from_errors(errors[4:], 4), 14)
    order, ps, code_bits, _ = orc.find_partitioned_rice_parameter(signal, 4, 14)
    assert code_bits <= single_bits
    assert order == 1
    assert ps[0] > ps[1]


def _table(vals):
    t = np.zeros(32, np.uint32)  # PrcBitTable::zero(), src/rice.rs:58-63
    t[: len(vals)] = vals
    return t


def test_partition_evaluation():
    """src/rice.rs:367-378 (eval_partitions)."""
    p1, b1 = orc.prc_minimizer(_table([17, 19, 15, 11, 19]), 4)
    p2, b2 = orc.prc_minimizer(_table([12, 14, 16, 18, 20]), 4)
    assert b1 + b2 == 23
    assert [p1, p2] == [3, 0]


def test_partition_merging():
    """src/rice.rs:380-391 (merge_partitions, offset 4)."""
    merged = orc.prc_merge(_table([17, 19, 15, 11, 19]), _table([12, 14, 16, 18, 20]), 4)
    assert merged[:5].tolist() == [25, 29, 27, 25, 35]
    # lanes where both inputs are 0 wrap to 2^32 - 4 and clamp to MAX_P_TO_BITS (rice.rs:147-150)
    assert merged[5] == orc.MAX_P_TO_BITS


def test_minimizer_search():
    """src/rice.rs:393-412 incl. the tie -> smallest p rule."""
    assert orc.prc_minimizer(_table([6, 7, 4, 5, 9, 0, 0, 0]), 4) == (2, 4)
    assert orc.prc_minimizer(_table([6, 7, 8, 5, 3, 0, 0, 0]), 4) == (4, 3)
    assert orc.prc_minimizer(_table([1, 7, 8, 5, 3, 0, 0, 0]), 4) == (0, 1)
    assert orc.prc_minimizer(_table([7, 1, 1, 1, 3, 0, 0, 0]), 4) == (1, 1)
    # high half (params 16..30), rice.rs:126-135
    t = np.full(32, 1000, np.uint32)
    t[20] = 7
    assert orc.prc_minimizer(t, 30) == (20, 7)
    assert orc.prc_minimizer(t, 15) == (0, 1000)


def test_prc_max_bits():
    """src/rice.rs:414-419: saturation at 2^27 - 1."""
    table = orc.prc_bit_table_from_errors([0x0FFFFFFE, 0x01000000], 0)
    assert table[0] == orc.MAX_P_TO_BITS


def test_signbit_coding():
    """src/rice.rs:169-187: 0, -1, 1, -2 -> 0, 1, 2, 3 and back."""
    assert [orc.encode_signbit(v) for v in (0, -1, 1, -2)] == [0, 1, 2, 3]
    for v in (0, -1, 1, -2, 12345, -12345, 2**31 - 1, -(2**31) + 1):
        assert orc.decode_signbit(orc.encode_signbit(v)) == v


# ------------------------------------------------- coding.rs / bitrepr.rs ----
def test_losslessness_residual_coding():
    """src/coding.rs:771-785."""
    signal = util.quantize(util.noise(3, 64, 0.4), 8)
    res = orc.encode_residual(signal, 0)
    assert (orc.decode_residual(res) == signal).all()
    signal = util.quantize(np.concatenate([util.noise(4, 2048, 0.9), util.sine(2048, 40, 0.1)]), 8)
    res = orc.encode_residual(signal, 0)
    assert (orc.decode_residual(res) == signal).all()


@pytest.mark.parametrize("case", ["noise", "sine"])
def test_losslessness_subframe_coding(case):
    """src/coding.rs:787-799 restricted to the LPC candidate (the path under test):
    estimated_qlpc output must decode (decode.rs:159-177) to the input."""
    bps = 8
    signal = (util.quantize(util.noise(8, 64, 0.4), bps) if case == "noise"
              else util.quantize(util.sine(64, 40, 0.9), bps))
    out = orc.estimated_qlpc(signal, bps, orc.make_config())
    assert out["status"] == 0
    res = dict(block_size=64, partition_order=out["rice_order"], rice_params=out["rice_params"],
               quotients=out["quotients"], remainders=out["remainders"])
    resid = orc.decode_residual(res)
    assert (resid == out["residual"]).all()
    dec = orc.decode_lpc(out["warm_up"], out["coefs"], out["shift"], resid)
    assert (dec == signal).all()


@pytest.mark.parametrize("seed,n,warm,max_p", [(1, 4096, 10, 30), (2, 4096, 1, 14), (3, 1152, 8, 30),
                                               (4, 64, 0, 30), (5, 8192, 24, 30)])
def test_residual_count_bits_is_accurate(seed, n, warm, max_p):
    """src/component/bitrepr.rs:706-717 ("`Residual::count_bits` should be accurate"):
    the formula (bitrepr.rs:533-544) equals the number of bits `write` emits."""
    errors = util.quantize(util.noise(seed, n, 0.3) * util.sine(n, 300, 1.0), 16)
    errors[:warm] = 0
    res = orc.encode_residual(errors, warm, max_p)
    assert res["count_bits"] == util.residual_write_bits(res)
    # search estimate vs written size differ only by the RICE2 parameter width
    nparts = 1 << res["partition_order"]
    rice2 = nparts if (res["rice_params"] > 14).any() else 0
    assert res["count_bits"] == 6 + res["code_bits"] + rice2
    assert (res["quotients"][:warm] == 0).all() and (res["remainders"][:warm] == 0).all()


def test_rice2_parameter_width():
    """bitrepr.rs:540-543: any p > 14 switches to 5-bit parameters."""
    errors = util.quantize(util.noise(9, 4096, 0.9), 24)
    res = orc.encode_residual(errors, 0, 30)
    assert (res["rice_params"] > 14).any()
    assert res["count_bits"] == util.residual_write_bits(res)
    res14 = orc.encode_residual(errors, 0, 14)
    assert (res14["rice_params"] <= 14).all()
    assert res14["count_bits"] == util.residual_write_bits(res14)


def test_lpc_count_bits_formula():
    """bitrepr.rs:492-499 on the doctest component of datatype.rs:2077-2084:
    Residual::new(0, 64, 1, &[8], zeros, zeros), QuantizedParameters::new(&[1], 1, 0, 7), bps 16."""
    rbits = orc.residual_count_bits(64, 1, 0, [8], 0, 8)
    assert rbits == 2 + 4 + 4 + 63 + (8 * 64 - 8)
    assert orc.lpc_count_bits(16, 1, 7, rbits) == 8 + 16 + 4 + 5 + 7 + rbits
    assert orc.verbatim_count_bits(4096, 16) == 8 + 4096 * 16  # datatype.rs:1944-1949


def test_midside_roundtrip():
    """coding.rs:476-484 vs decode.rs:91-103."""
    l = util.quantize(util.noise(21, 512, 0.9), 16)
    r = util.quantize(util.noise(22, 512, 0.9), 16)
    m, s = orc.stereo_to_midside(l, r)
    assert (m == ((l.astype(np.int64) + r) >> 1)).all() and (s == l - r).all()
    l2, r2 = orc.midside_to_stereo(m, s)
    assert (l2 == l).all() and (r2 == r).all()


# --------------------------------------------------- whole path properties ----
@pytest.mark.parametrize("n,bps,order,window", [
    (4096, 16, 8, ("tukey", 0.4)), (4096, 16, 10, ("tukey", 0.4)), (4096, 17, 10, ("tukey", 0.4)),
    (8192, 24, 24, ("tukey", 0.4)), (8192, 25, 32, ("tukey", 0.4)), (16384, 24, 24, ("tukey", 0.4)),
    (1152, 16, 12, ("tukey", 0.5)), (576, 8, 6, "rectangle"), (1001, 16, 10, ("tukey", 0.4)),
    (64, 16, 10, ("tukey", 0.4)), (32767, 16, 8, ("tukey", 0.1)),
])
def test_estimated_qlpc_is_lossless(n, bps, order, window):
    """fuzz/fuzz_targets/frame_encode.rs:197-212 property on the path: decode == input,
    and Residual invariants of src/component/verify.rs:274-332."""
    signal = util.sine_noise(n, bps, 36, 0.4, 0.04, seed=n + order)
    cfg = orc.make_config(lpc_order=order, window=window)
    out = orc.estimated_qlpc(signal, bps, cfg)
    assert out["status"] == 0
    assert 1 <= out["order"] <= order
    assert 0 <= out["shift"] <= 15
    assert (np.abs(out["coefs"].astype(np.int32)) <= (1 << 14)).all()
    nparts = 1 << out["rice_order"]
    assert n % nparts == 0 and (n >> out["rice_order"]) >= out["order"]
    assert (out["rice_params"] <= 30).all()
    assert (out["residual"][: out["order"]] == 0).all()
    dec = orc.decode_lpc(out["warm_up"], out["coefs"], out["shift"], out["residual"])
    assert (dec == signal).all()
    res = dict(block_size=n, warmup_length=out["order"], partition_order=out["rice_order"],
               rice_params=out["rice_params"], quotients=out["quotients"])
    assert out["residual_bits"] == util.residual_write_bits(res)


def test_canonical_vs_reference_order_tolerance():
    """T2 of the parity contract: the build's canonical summation order agrees with the
    reference (nosimd) order within 1e-12 relative on R and 1e-8 relative on LPC
    coefficients (same bound class as the reference's own simd/nosimd parity test,
    src/lpc.rs:1392-1413, which only asserts rtol 1e-5)."""
    worst_r, worst_a, same, total = 0.0, 0.0, 0, 0
    for seed in range(24):
        bps = 16 if seed % 2 == 0 else 24
        n = [4096, 8192, 1152][seed % 3]
        signal = util.sine_noise(n, bps, 20 + 7 * seed, 0.5, 0.02 + 0.02 * (seed % 5), seed=seed)
        ref = orc.estimated_qlpc(signal, bps, orc.make_config(lpc_order=10))
        can = orc.estimated_qlpc(signal, bps, orc.make_config(lpc_order=10, acorr=orc.ACORR_CANONICAL))
        worst_r = max(worst_r, float(np.max(np.abs(ref["autocorr"] - can["autocorr"])
                                            / np.abs(ref["autocorr"][0]))))
        worst_a = max(worst_a, float(np.max(np.abs(ref["lpc_coefs"] - can["lpc_coefs"])
                                            / np.max(np.abs(ref["lpc_coefs"])))))
        total += 1
        if (ref["coefs"].tolist(), ref["shift"]) == (can["coefs"].tolist(), can["shift"]):
            same += 1
            assert (ref["residual"] == can["residual"]).all()
            assert ref["rice_params"].tolist() == can["rice_params"].tolist()
            assert ref["subframe_bits"] == can["subframe_bits"]
    assert worst_r < 1e-12, worst_r
    assert worst_a < 1e-8, worst_a
    assert same >= total - 1, (same, total)


def test_nightly_simd_order_known_samples():
    """src/lpc.rs:1024-1041 again, through the simd-nightly summation order (lpc.rs:510-531):
    integer data -> exact, whatever the alignment of the buffer."""
    for base_mod in (0, 3, 16, 37):
        corr = orc.auto_correlation_nightly(33, np.array(KNOWN, np.float32), base_mod)
        assert (corr[0], corr[1], corr[2], corr[32]) == (24.0, -4.0, 2.0, 0.0)


def test_parity_of_auto_correlation_functions_for_simd_and_nosimd():
    """src/lpc.rs:1392-1413: Sine(32, 0.8) + noise(0.01), 16-bit, 1024 samples, order 25;
    the reference asserts assert_close (rtol 1e-5) between its two orders."""
    signal = util.sine_noise(1024, 16, 32, 0.8, 0.01, seed=77).astype(np.float32)
    a = orc.auto_correlation_nightly(25, signal)
    b = orc.auto_correlation(25, signal)
    for x, y in zip(a, b):
        assert_close(float(x), float(y))


def test_three_summation_orders_agree_equally_well():
    """The reference has two summation orders of its own (stable `nosimd` lpc.rs:533-548 and
    `simd-nightly` lpc.rs:510-531, the one its published numbers use) that differ in the last
    bits (compression ratio 0.52764995 vs 0.52764889, report/report.{stable,nightly}.md:16).
    The build's canonical order must sit inside that spread: every pairwise distance is at the
    1e-13 level relative to R[0], and the quantised coefficients agree just as often."""
    worst = {"stable-nightly": 0.0, "stable-canonical": 0.0, "nightly-canonical": 0.0}
    same = {"stable-nightly": 0, "stable-canonical": 0, "nightly-canonical": 0}
    total = 0
    for seed in range(40):
        bps = 16 if seed % 2 == 0 else 24
        n = [4096, 4608, 8192, 1152][seed % 4]
        signal = util.sine_noise(n, bps, 17 + 5 * seed, 0.5, 0.01 + 0.03 * (seed % 7), seed=3000 + seed)
        res = {name: orc.estimated_qlpc(signal, bps, orc.make_config(lpc_order=12, acorr=mode))
               for name, mode in (("stable", orc.ACORR_REFERENCE), ("nightly", orc.ACORR_NIGHTLY),
                                  ("canonical", orc.ACORR_CANONICAL))}
        total += 1
        for key in worst:
            a, b = (res[k] for k in key.split("-"))
            d = float(np.max(np.abs(a["autocorr"] - b["autocorr"])) / abs(a["autocorr"][0]))
            worst[key] = max(worst[key], d)
            if (a["coefs"].tolist(), a["shift"]) == (b["coefs"].tolist(), b["shift"]):
                same[key] += 1
                assert np.array_equal(a["residual"], b["residual"])
                assert a["subframe_bits"] == b["subframe_bits"]
            for r in (a, b):
                dec = orc.decode_lpc(r["warm_up"], r["coefs"], r["shift"], r["residual"])
                assert np.array_equal(dec, signal)
    print("\nmax |dR|/R0:", worst, " identical quantised coefficients:", same, "of", total)
    assert max(worst.values()) < 1e-12
    assert worst["stable-canonical"] < 10 * max(worst["stable-nightly"], 1e-16) + 1e-15
    assert min(same.values()) >= total - 2


# ------------------------------------------------- fixed LPC (coding.rs) ----
def test_fixed_lpc_error_computation():
    """src/coding.rs:707-723: Sine(32, 0.3) + noise(0.1), 16-bit, 64 samples."""
    signal = util.sine_noise(64, 16, 32, 0.3, 0.1, seed=9)
    errors = orc.reset_fixed_lpc_errors(signal)
    s = signal.astype(np.int64)
    assert np.array_equal(errors[0], signal)
    assert np.array_equal(errors[1][1:], (s[1:] - s[:-1]).astype(np.int32))
    assert np.array_equal(errors[2][2:], (s[2:] - 2 * s[1:-1] + s[:-2]).astype(np.int32))
    # carry starts at 0 (coding.rs:188): the first k entries are partial differences
    assert errors[1][0] == signal[0] and errors[2][0] == signal[0]
    assert errors[2][1] == signal[1] - 2 * signal[0]


def test_fixed_lpc_of_sine():
    """src/coding.rs:725-737: Sine(100, 0.6), 8-bit, 1024 samples; max_order 0..4 all decode."""
    signal = util.quantize(util.sine(1024, 100, 0.6), 8)
    for max_order in range(5):
        sub = orc.fixed_lpc(signal, 8, 2 ** 64 - 1, orc.make_fixed_config(max_order=max_order))
        assert sub["selected"] and sub["order"] <= max_order
        assert np.array_equal(orc.decode_fixed(sub["warm_up"], sub["residual"]), signal)
        # FixedLpc::count_bits, bitrepr.rs:473-477, against the bits Residual::write emits
        assert sub["subframe_bits"] == 8 + 8 * sub["order"] + sub["residual_bits"]


def test_order_selector_bitcount():
    """src/coding.rs:944-980: errors 255 / 256 / 128 (orders 0, 1, 2), 256 samples, 16-bit ->
    order 0, and no order has fewer real bits."""
    errs = [np.full(256, v, np.int32) for v in (255, 256, 128)]
    keys = [16 * k + orc.find_partitioned_rice_parameter(e, k, 30)[2] for k, e in enumerate(errs)]
    assert int(np.argmin(keys)) == 0
    counts = []
    for k, e in enumerate(errs):
        res = orc.encode_residual(e, k)
        counts.append(orc.residual_count_bits(256, k, res["partition_order"], res["rice_params"],
                                              res["sum_quotients"], res["sum_rice_params"]) + 16 * k)
    assert all(c >= counts[0] for c in counts)


def test_order_selector_approxent():
    """src/coding.rs:982-1004: errors 255 / 256 / 128 / 127, ApproxEnt{partitions: 32} -> order 2."""
    errs = [np.full(256, v, np.int32) for v in (255, 256, 128, 127)]
    for mode in (orc.SUMABS_STABLE, orc.SUMABS_NIGHTLY, orc.SUMABS_CANONICAL):
        keys = [orc.estimate_entropy(e, k, 32, mode) + 16 * k for k, e in enumerate(errs)]
        assert int(np.argmin(keys)) == 2  # min_by_key: first minimum


def test_log2f_restatement_matches_host_libm():
    """f32::log2 -> libm log2f (glibc 2.35 here).  The restatement shared by oracle and kernel
    must equal it bit for bit; a strided sweep over every binade of the positive floats plus the
    arguments the estimator really uses (1/(avg+1) and 1 - 1/(avg+1))."""
    import ctypes
    libm = ctypes.CDLL("libm.so.6")
    libm.log2f.argtypes = [ctypes.c_float]
    libm.log2f.restype = ctypes.c_float
    bits = np.arange(0x00000001, 0x7f800000, 104729, dtype=np.uint32)
    avg = np.float32(2.0) * np.arange(1, 200000, 37, dtype=np.float32) / np.float32(256.00001)
    gp = (np.float32(1.0) / (avg + np.float32(1.0))).astype(np.float32)
    xs = np.concatenate([bits.view(np.float32), gp, (np.float32(1.0) - gp).astype(np.float32),
                         np.array([1.0, 0.5, 2.0, 0.0, np.inf], np.float32)])
    for x in xs:
        a, b = orc.log2f(float(x)), libm.log2f(float(x))
        assert a == b or (np.isnan(a) and np.isnan(b)), (float(x).hex(), a, b)
    assert np.isnan(orc.log2f(-1.0)) and orc.log2f(0.0) == -np.inf


def test_sum_abs_orders_agree_below_2_24():
    """find_sum_abs_f32 (arrayutils.rs:496-506): while every partial sum stays below 2^24 the
    stable chain, the simd-nightly lanes and the exact-integer canonical definition are the same
    number; beyond it they differ by f32 rounding only, and the canonical one is the correctly
    rounded sum."""
    rng = np.random.default_rng(5)
    small = rng.integers(-2 ** 14, 2 ** 14, 256).astype(np.int32)  # sum <= 2^22
    vals = {orc.find_sum_abs_f32(small, m, off) for m in range(3) for off in (0, 5, 16)}
    assert len(vals) == 1 and vals.pop() == float(np.abs(small.astype(np.int64)).sum())
    big = rng.integers(-2 ** 22, 2 ** 22, 1024).astype(np.int32)   # sum ~ 2^31: rounding shows
    exact = int(np.abs(big.astype(np.int64)).sum())
    can = orc.find_sum_abs_f32(big, orc.SUMABS_CANONICAL)
    assert can == float(np.float32(exact))
    for m in (orc.SUMABS_STABLE, orc.SUMABS_NIGHTLY):
        assert abs(orc.find_sum_abs_f32(big, m) - exact) <= 1e-5 * exact


def test_encode_subframe_with_fixed_candidate():
    """encode_subframe, src/coding.rs:384-418: LPC must beat min(verbatim, fixed) strictly;
    otherwise fixed if it beats verbatim; the result always decodes."""
    fc = orc.make_frame_config(orc.make_config(lpc_order=8))
    kinds = set()
    for seed, (amp, namp, period) in enumerate([(0.4, 0.4, 200), (0.6, 0.0, 100), (0.3, 0.001, 37),
                                                (0.0, 0.9, 50), (0.2, 0.02, 16)]):
        x = util.sine_noise(4096, 16, period, amp, namp, seed=40 + seed)
        r = orc.encode_subframe(x, 16, fc)
        kinds.add(r["kind"])
        verbatim = orc.verbatim_count_bits(4096, 16)
        assert r["bits"] <= verbatim
        fx = orc.fixed_lpc(x, 16, verbatim)
        lp = orc.estimated_qlpc(x, 16, fc.qlpc)
        if r["kind"] == orc.KIND_LPC:
            assert lp["subframe_bits"] < min(verbatim, fx["subframe_bits"] if fx["selected"] else verbatim)
            dec = orc.decode_lpc(lp["warm_up"], lp["coefs"], lp["shift"], r["residual"])
        elif r["kind"] == orc.KIND_FIXED:
            assert fx["selected"] and fx["subframe_bits"] < verbatim and lp["subframe_bits"] >= fx["subframe_bits"]
            dec = orc.decode_fixed(x[:fx["order"]], r["residual"])
        else:
            continue
        assert np.array_equal(dec, x)
    assert orc.KIND_LPC in kinds


# ------------------------------------------------- bit writer (bitrepr.rs) ----
def test_utf8_encoding():
    """src/component.rs:59-77."""
    assert orc.encode_to_utf8like(0x56) == bytes([0x56])
    assert orc.encode_to_utf8like(0x1024) == bytes([0xE1, 0x80, 0xA4])
    assert orc.encode_to_utf8like(0xFFFFFFFFF) == bytes([0xFE, 0xBF, 0xBF, 0xBF, 0xBF, 0xBF, 0xBF])
    assert orc.encode_to_utf8like(0x1000000000) is None  # out of domain


def test_crc_catalogue_check_values():
    """crc::CRC_8_SMBUS / crc::CRC_16_UMTS (bitrepr.rs:39-40): the catalogue's check values for
    b"123456789"."""
    assert orc.crc8(b"123456789") == 0xF4
    assert orc.crc16(b"123456789") == 0xFEE8


def test_write_frame_header():
    """src/component/bitrepr.rs:635-667: block 192, Independent(2), Unspecified specs, variable
    blocking, start sample 0 -> the bit string incl. CRC-8 0x69; and the doctest of FrameHeader::new
    (datatype.rs:1588-1599): 192 / mono / 8-bit / 44.1 kHz / StartSample(123456)."""
    h = orc.write_frame_header(192, 1, 0, 0, True, 0)
    assert h == bytes([0b11111111, 0b11111001, 0b00010000, 0b00010000, 0b00000000, 0b01101001])
    h = orc.write_frame_header(192, 0, 8, 44100, True, 123456)
    assert h[:8] == bytes([0xFF, 0xF9, 0x19, 0x02, 0xF0, 0x9E, 0x89, 0x80])
    # FrameHeader::count_bits (bitrepr.rs:361-371): 40 + utf8 + extra block-size / rate bits
    assert len(orc.write_frame_header(2304, 1, 16, 44100, False, 0)) == 6
    assert len(orc.write_frame_header(1000, 1, 16, 12345, False, 0x1024)) == 5 + 3 + 2 + 2


def test_subframe_byte_layouts():
    """Doctests of Constant / Verbatim / FixedLpc / Lpc::new, src/component/datatype.rs:1839-1845,
    1914-1923, 1986-1991, 2077-2084, and verify_bit_counter (count_bits == bits written)."""
    b, bits = orc.write_subframe(0, 16, 1024, dc_offset=3)
    assert b == bytes([0x00, 0x00, 0x03]) and bits == 8 + 16
    b, bits = orc.write_subframe(1, 16, 64, samples=np.full(64, 0xAB))
    assert b[0] == 0x02 and all(b[1 + 2 * t:3 + 2 * t] == bytes([0x00, 0xAB]) for t in range(64))
    assert bits == orc.verbatim_count_bits(64, 16)
    zeros = dict(rice_order=0, rice_params=[8], residual=np.zeros(64))
    b, bits = orc.write_subframe(2, 16, 64, samples=[0xCD] + [0] * 63, order=1, **zeros)
    assert b[0] == 0x12 and b[1:3] == bytes([0x00, 0xCD])
    b, bits = orc.write_subframe(3, 16, 64, samples=[0xEF] + [0] * 63, order=1, shift=0, precision=7,
                                 coefs=[1] + [0] * 31, **zeros)
    assert b[0] == 0x40 and b[1:3] == bytes([0x00, 0xEF]) and b[3:5] == bytes([0x60, 0x01])
    res_bits = orc.residual_count_bits(64, 1, 0, np.array([8], np.uint8), 0, 8)
    assert bits == orc.lpc_count_bits(16, 1, 7, res_bits)


def test_written_frames_parse_back_to_the_input():
    """Frame::write (bitrepr.rs:289-319) for every SubFrame kind and channel assignment: the bytes
    decode to the input with an independent FLAC parser (tests/flac_parse.py), both CRCs match, and
    the length equals Frame::count_bits (bitrepr.rs:275-287)."""
    import flac_parse
    n, bps = 4096, 16
    t = np.arange(n)
    frames = [np.stack([util.sine_noise(n, bps, 200, 0.4, 0.05, seed=1), util.sine_noise(n, bps, 170, 0.3, 0.05, seed=2)]),
              np.stack([t // 7, t // 5 + 3]),
              np.stack([np.full(n, 1234), util.quantize(util.noise(3, n, 0.999), bps)]),
              np.stack([util.sine_noise(n, bps, 90, 0.5, 0.01, seed=4)] * 2)]
    frames[3][1] = frames[3][0] * 7 // 8
    x = np.stack(frames).astype(np.int32)
    fc = orc.make_frame_config(orc.make_config(lpc_order=8))
    res, resid = orc.encode_stereo_frames_cfg(x, bps, fc)
    seen = set()
    for f in range(len(x)):
        data = orc.write_stereo_frame(res[f], x[f, 0], x[f, 1], bps, 44100, 1000 + f, resid[f, 0], resid[f, 1])
        got = flac_parse.parse_frame(data)
        assert got["length"] == len(data) and got["number"] == 1000 + f and not got["variable"]
        assert got["block_size"] == n and got["sample_rate"] == 44100 and got["bps"] == bps
        assert np.array_equal(got["channels"], x[f])
        roles = res[f]["role"]
        sub_bits = sum(int(res[f]["bits"][r]) for r in roles)
        assert len(data) * 8 == (8 * 7 + sub_bits + 7) // 8 * 8 + 16   # header here is 7 bytes
        seen.update(got["kinds"])
    assert seen == {"constant", "verbatim", "fixed", "lpc"}


# ------------------------------------------------- input side (arrayutils.rs) ----
def test_do_deinterleave():
    """src/arrayutils.rs:671-686."""
    inter = [0, 0, -1, -2, 1, 2, -3, 6]
    assert orc.deinterleave(inter, 2, 4).tolist() == [0, -1, 1, -3, 0, -2, 2, 6]
    assert orc.deinterleave(inter, 4, 3, dest_len=12, fill=-123).tolist() == [0, 1, 0, 0, 2, 0, -1, -3, 0, -2, 6, 0]


def test_convert_le_bytes_to_ints():
    """src/arrayutils.rs:711-728 (3-byte and 1-byte samples)."""
    b = bytes([0x56, 0x34, 0x12, 0x9B, 0x57, 0x13, 0xFF, 0xFF, 0xFF, 0xAC, 0x68, 0x24])
    assert orc.le_bytes_to_i32s(b, 3).tolist() == [0x123456, 0x13579B, -1, 0x2468AC]
    b = bytes([0x56, 0x34, 0x12, 0x9B, 0x80, 0x13, 0xFF, 0x68])
    assert orc.le_bytes_to_i32s(b, 1).tolist() == [0x56, 0x34, 0x12, -0x65, -0x80, 0x13, -0x01, 0x68]
    # the Fill doctest, src/source.rs:72-80
    assert orc.deinterleave(orc.le_bytes_to_i32s(bytes([0x12, 0x34, 0x54, 0x76, 0x56, 0x78, 0x10, 0x32]), 2),
                            2, 2).tolist() == [0x3412, 0x7856, 0x7654, 0x3210]


# ---- experimental estimators: the reference's own tests of lpc.rs, restated ---------------------
# The solver is nalgebra's (not in the reference tree): these pin the oracle as far as the reference pins itself.
def test_lagged_outer_prod_sum_computation():
    """src/lpc.rs:1339-1359, exact values"""
    m = orc.lagged_outer_prod_sum(2, [4.0, -4.0, 3.0, -3.0, 2.0, -2.0, 1.0, -1.0])
    assert m[0, 0] == float(-4 * -4 + 3 * 3 + -3 * -3 + 2 * 2 + -2 * -2 + 1 * 1 + -1 * -1)
    assert m[0, 1] == float(4 * -4 + -4 * 3 + 3 * -3 + -3 * 2 + 2 * -2 + -2 * 1 + 1 * -1)
    assert m[1, 1] == float(4 * 4 + -4 * -4 + 3 * 3 + -3 * -3 + 2 * 2 + -2 * -2 + 1 * 1)
    assert m[1, 0] == m[0, 1]


def test_lpc_with_known_coefs_dmse():
    """src/lpc.rs:1194-1212: direct MSE recovers the generating filter (1, -1, 0.5)"""
    signal = [0, -512, 0, 512, 256, -256, -256, 128, 256, 0, -192, -64, 128, 96, -64, -96, 16, 80, 16, -56, -32, 32,
              36, -12]
    cfg = orc.make_config(lpc_order=3, window="rectangle", use_direct_mse=True)
    _, _, coefs, st = orc.lpc_with_direct_mse(signal, cfg)
    assert st == 0
    assert 0.9 < coefs[0] < 1.1 and -1.1 < coefs[1] < -0.9 and 0.4 < coefs[2] < 0.6


def test_solve_mut_sym():
    """src/lpc.rs:1361-1388: covar * solve(covar, autocorr[1..]) == autocorr[1..] (assert_close: rtol 1e-5)"""
    s = util.quantize(util.sine(1024, 32, 0.8) + util.noise(3, 1024, 0.01), 16).astype(np.float32)
    order = 12
    R = orc.auto_correlation(order + 1, s)
    G = orc.lagged_outer_prod_sum(order, s)
    ok, x = orc.cholesky_solve(G, R[1:])
    assert ok
    np.testing.assert_allclose(G @ x, R[1:], rtol=1e-5)
    np.testing.assert_allclose(x, np.linalg.solve(G, R[1:]), rtol=1e-6)  # and it is the solution LAPACK finds


def test_cholesky_solver_against_third_parties():
    """X1's solver is nalgebra 0.32's Cholesky, restated from its published algorithm (the crate is not under
    /root/reference): no reference-produced vector can pin it.  This bounds it by third parties instead: (a) LAPACK
    (numpy's dpotrf / dpotrs route through np.linalg.cholesky / solve_triangular-free algebra) on well- and
    ill-conditioned Gram matrices of every order 1..32 to 1e-12 of the conditioning-scaled solution, (b) exact small-integer
    systems whose factors and solutions are integers or dyadic rationals -- every operation of either algorithm is exact
    there, so the restatement must return them to the last bit."""
    rng = np.random.default_rng(20251004)
    for order in range(1, 33):
        for trial in range(4):
            n = 256 + 64 * trial
            x = util.quantize(util.sine(n + order, 23.0 + order + 3 * trial, 0.5) + util.noise(order * 7 + trial, n + order, 0.2 / (1 + 3 * trial)),
                              16).astype(np.float64)
            X = np.stack([x[order - 1 - i:order - 1 - i + n] for i in range(order)])  # lagged vectors
            G = X @ X.T
            b = X @ x[order:order + n]
            ok, got = orc.cholesky_solve(G, b)
            assert ok
            L = np.linalg.cholesky(G)  # LAPACK dpotrf
            want = np.linalg.solve(L.T, np.linalg.solve(L, b))  # the two triangular solves (dtrtrs)
            cond = np.linalg.cond(G)
            tol = 1e-12 * max(1.0, cond / 1e3)  # forward error of a backward-stable solve ~ cond * 2^-53
            assert np.max(np.abs(got - want)) <= tol * np.max(np.abs(want)), (order, trial, cond)
    # exact cases: G = L L^T with small-integer lower-triangular L and power-of-two diagonals, b = G z for integer z
    for order in (1, 2, 3, 5, 8, 12, 24, 32):
        for trial in range(6):
            L = np.tril(rng.integers(-3, 4, (order, order))).astype(np.float64)
            np.fill_diagonal(L, 2.0 ** rng.integers(0, 3, order))
            G = L @ L.T
            z = rng.integers(-5, 6, order).astype(np.float64)
            ok, got = orc.cholesky_solve(G, G @ z)
            assert ok
            assert np.array_equal(got, z), (order, trial)


def test_cholesky_rejects_what_nalgebra_rejects():
    """Cholesky::new_internal: a zero or negative pivot -> None -> the caller's regulariser loop (lpc.rs:887-896)"""
    assert orc.cholesky_solve(np.zeros((3, 3)), np.ones(3))[0] is False
    assert orc.cholesky_solve(np.array([[1.0, 2.0], [2.0, 1.0]]), np.ones(2))[0] is False
    ok, x = orc.cholesky_solve(np.array([[4.0, 2.0], [2.0, 3.0]]), np.array([2.0, 1.0]))
    assert ok and np.allclose(x, [0.5, 0.0])
    # all-zero block: Gram = 0 -> regulariser 1 -> coefficients 0
    _, _, coefs, st = orc.lpc_with_direct_mse(np.zeros(256, np.int32), orc.make_config(lpc_order=8, use_direct_mse=True))
    assert st == 0 and not coefs.any()


def test_if_direct_mse_is_better_than_autocorr():
    """src/lpc.rs:1297-1337: sus109 ch0, 128 samples, order 24"""
    sg = util.test_signal("sus109", 0)[:128]
    ca = orc.lpc_from_autocorr(sg, orc.make_config(lpc_order=24, window=("tukey", 0.1)))[1]
    cd = orc.lpc_with_direct_mse(sg, orc.make_config(lpc_order=24, window="rectangle", use_direct_mse=True))[2]
    energy = (sg.astype(np.float64) ** 2).sum()
    ea = orc.compute_raw_errors(sg, ca)[24:].astype(np.float64)
    ed = orc.compute_raw_errors(sg, cd)[24:].astype(np.float64)
    assert 10 * np.log10(energy / (ea ** 2).sum()) < 10 * np.log10(energy / (ed ** 2).sum())


@pytest.mark.parametrize("block_size", [256, 512, 1024, 2048, 4096])
def test_comparing_mse_vs_mae(block_size):
    """src/lpc.rs:1448-1486: IRLS (4 steps) does not increase the mean absolute error"""
    sg = util.test_signal("sus109", 0)[:block_size]
    cfg = orc.make_config(lpc_order=16, window="rectangle", use_direct_mse=True)
    c_mse = orc.lpc_with_direct_mse(sg, cfg)[2]
    c_mae, st = orc.lpc_with_irls_mae(sg, cfg, 4)
    assert st == 0
    mae_mse = (np.abs(orc.compute_raw_errors(sg, c_mse)) / np.float32(len(sg))).sum(dtype=np.float32)
    mae_mae = (np.abs(orc.compute_raw_errors(sg, c_mae)) / np.float32(len(sg))).sum(dtype=np.float32)
    assert mae_mse >= mae_mae


def test_perform_qlpc_switches():
    """perform_qlpc (src/coding.rs:333-351): steps are ignored without use_direct_mse; with it the subframe is
    still lossless"""
    x = util.sine_noise(4096, 16, 57, 0.5, 0.05, seed=3)
    a = orc.estimated_qlpc(x, 16, orc.make_config(lpc_order=10))
    b = orc.estimated_qlpc(x, 16, orc.make_config(lpc_order=10, use_direct_mse=True))
    c = orc.estimated_qlpc(x, 16, orc.make_config(lpc_order=10, use_direct_mse=True, mae_optimization_steps=2))
    for r in (a, b, c):
        k = r["order"]
        assert np.array_equal(orc.decode_lpc(x[:k], r["coefs"], r["shift"], r["residual"]), x)
    # (on a long stationary block under a Tukey window the two estimators agree to 1e-8 and quantise alike;
    # a short block shows the difference)
    y = x[:200]
    a = orc.estimated_qlpc(y, 16, orc.make_config(lpc_order=10, window="rectangle"))
    b = orc.estimated_qlpc(y, 16, orc.make_config(lpc_order=10, window="rectangle", use_direct_mse=True))
    assert b["coefs"].tolist() != a["coefs"].tolist()
