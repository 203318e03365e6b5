"""Soundness of the order certificate on the CPU, through the oracle's statement of it
(orc_quant_certified == the kernel's arithmetic, operation for operation; tests/test_gpu_certified_order.py holds the GPU to
the oracle bit for bit, counters included).  The certificate is a first-order perturbation bound with a safety factor, not
a theorem about the floating-point recursion (oracle/flacenc_oracle.c, orc_quant_certified, lists what is shown and what
is assumed): this is its evidence -- over tens of thousands of subframes drawn to stress
it (near-pure and multi-tone material with Toeplitz condition numbers up to 1e9, every order 1..12, precisions 3..15, all
windows, 8..24 bits, silence / constants / impulses / clipping) a CERTIFIED subframe's QuantizedParameters never differ
from the reference order's, while the bare kernel order's do on some of the very same subframes."""
import numpy as np
import pytest

import util
from oracle import oracle as orc


def _corpus(rng, count, n, bps):
    t = np.arange(n, dtype=np.float64)
    full = float(1 << (bps - 1))
    rows = []
    for i in range(count):
        kind = i % 8
        if kind == 0:  # near-pure sine, noise floor 60..100 dB down
            x = 0.6 * np.sin(2 * np.pi * t / rng.uniform(6.0, 400.0) + rng.uniform(0, 6.28)) + rng.uniform(-1, 1, n) * 10 ** rng.uniform(-5, -3)
        elif kind == 1:  # two close tones
            p = rng.uniform(8.0, 120.0)
            x = 0.4 * np.sin(2 * np.pi * t / p) + 0.4 * np.sin(2 * np.pi * t / (p * rng.uniform(1.001, 1.05))) + rng.uniform(-1, 1, n) * 1e-4
        elif kind == 2:  # sine + noise (the bench family)
            x = 0.4 * np.sin(2 * np.pi * t / rng.uniform(20, 300)) + rng.uniform(-1, 1, n) * rng.uniform(0.01, 0.4)
        elif kind == 3:  # decaying resonance (strongly predictable)
            x = 0.9 * np.exp(-t / rng.uniform(500, 5000)) * np.sin(2 * np.pi * t / rng.uniform(10, 60))
        elif kind == 4:  # clipped tone
            x = np.clip(1.7 * np.sin(2 * np.pi * t / rng.uniform(30, 200)), -0.999, 0.999)
        elif kind == 5:  # DC + impulse(s), one possibly in front of t = P
            x = np.full(n, rng.uniform(-0.5, 0.5)) + rng.uniform(-1, 1, n) * 1e-4
            x[rng.integers(0, n)] = 0.99
            x[rng.integers(0, 12)] = -0.99
        elif kind == 6:  # low-pass noise (AR(1) close to the unit circle)
            e = rng.uniform(-1, 1, n) * 0.01
            x = np.zeros(n)
            a = rng.uniform(0.95, 0.9999)
            for k in range(1, n):
                x[k] = a * x[k - 1] + e[k]
            x *= 0.8 / max(1e-9, np.abs(x).max())
        else:  # square wave with jitter
            x = np.where(((t + rng.integers(0, 50)) // rng.integers(5, 80)) % 2 == 0, 0.8, -0.8) + rng.uniform(-1, 1, n) * 1e-3
        rows.append(np.clip(np.rint(x * full), -full, full - 1).astype(np.int32))
    rows.append(np.zeros(n, np.int32))
    rows.append(np.full(n, 17, np.int32))
    return np.stack(rows)


@pytest.mark.parametrize("n", [4096, 4608])
def test_a_certified_subframe_is_never_wrong(n):
    rng = np.random.default_rng(20251004 + n)
    total = recomputed = tier2 = tree_differs = 0
    for rnd in range(24 if n == 4096 else 8):
        order = int(rng.integers(1, 13))
        precision = int(rng.integers(3, 16))
        bps = int(rng.choice([8, 12, 16, 16, 16, 20, 24]))
        window = [("tukey", 0.4), ("tukey", 0.1), ("tukey", 1.0), "rectangle"][rnd % 4]
        x = _corpus(rng, 160, n, bps)
        kw = dict(lpc_order=order, quant_precision=precision, window=window)
        orc.cert_stats(reset=True)
        cp, cres, _, _ = orc.qlpc_batch(x, bps, orc.make_config(acorr=orc.ACORR_CANONICAL, **kw), nthreads=1, want_fp=False)
        st = orc.cert_stats()
        rp, rres, _, _ = orc.qlpc_batch(x, bps, orc.make_config(acorr=orc.ACORR_REFERENCE, **kw), want_fp=False)
        tp, _, _, _ = orc.qlpc_batch(x, bps, orc.make_config(acorr=orc.ACORR_CHUNK_TREE, **kw), want_fp=False)
        for f in ("coefs", "shift", "order", "status", "rice_order", "code_bits", "subframe_bits"):
            assert np.array_equal(cp[f], rp[f]), (n, order, precision, bps, window, f)
        assert np.array_equal(cres, rres)
        total += st[0]
        tier2 += st[1]
        recomputed += st[2]
        tree_differs += int(((tp["coefs"] != rp["coefs"]).any(axis=1) | (tp["shift"] != rp["shift"])).sum())
    print(f"n = {n}: {total} subframes, {tier2} needed the rows of T^-1, {recomputed} recomputed from the reference's chains "
          f"({recomputed / total:.3f}), 0 certified-but-different; the bare kernel order differs in {tree_differs}")
    assert total >= 1000 and recomputed < total  # (the certificate does certify most of even this corpus)


def test_a_system_that_is_not_positive_definite_is_not_certified():
    """Round 6's counter-example to the round-5 rule (tests/golden/README.md): the block opens on a clipped plateau, the
    Toeplitz matrix of its lag sums has a negative eigenvalue, the recursion is unstable on it and its two runs -- on the
    reference's sums and on the kernel's -- end 3.1e-7 apart where the bound (2 F_i = 4.6e-9) had certified the subframe.
    The rule now excludes systems with a non-positive denominator: the subframe is recomputed from the reference's chains."""
    import os
    x = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cert_nonpd_plateau_24bit.npy"))
    kw = dict(lpc_order=8, quant_precision=15, window="rectangle")
    orc.cert_stats(reset=True)
    cp, cres, _, _ = orc.qlpc_batch(x[None, :], 24, orc.make_config(acorr=orc.ACORR_CANONICAL, **kw), nthreads=1)
    assert orc.cert_stats() == (1, 0, 1), "analysed, no second tier, recomputed"
    assert orc.certificate_bounds(x, orc.make_config(**kw)) is None
    rp, rres, rR, ra = orc.qlpc_batch(x[None, :], 24, orc.make_config(acorr=orc.ACORR_REFERENCE, **kw))
    _, _, kR, ka = orc.qlpc_batch(x[None, :], 24, orc.make_config(acorr=orc.ACORR_CHUNK_TREE, **kw))
    assert np.array_equal(cp, rp) and np.array_equal(cres, rres)
    R = kR[0, :9]
    T = np.array([[R[abs(i - j)] for j in range(8)] for i in range(8)])
    assert np.linalg.eigvalsh(T).min() < 0, "the fixture's point: lag sums from t = P on need not be an autocorrelation"
    assert np.abs(ra[0, :8] - ka[0, :8]).max() > 1e-7  # (the round-5 bound for this subframe was 4.6e-9)


def test_attack_on_the_certificate_stays_an_order_of_magnitude_below_the_bound():
    """tools/certificate_attack.py for a bounded time: a hill-climber over signal parameters that maximises
    |a^_ref - a^_kernel|_i / bound_i among certifiable subframes (every bound below half a quantisation step).  Four runs of
    170 s found 0.030-0.040 after round 6's exclusion (1.5 ... 68 before it); VERDICT r5 asks for <= 0.1.  Also: the residual
    constant c_L of the floating-point recursion that the stated bound assumes to be <= 11."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import certificate_attack as ca
    worst, c_l, evals = ca.attack(25.0, seed=20251005, orders=[4, 8, 10, 12], log=lambda *_: None)
    print(f"{evals} evaluations: worst actual / bound {worst[0]:.4f}, largest recursion constant c_L {c_l:.3f}")
    assert evals > 2000
    assert worst[0] <= 0.1, worst
    assert c_l <= 2.0, c_l


@pytest.mark.parametrize("n,order", [(1000, 8), (1152, 12), (256, 4), (8192, 10), (16384, 15), (20000, 6)])
def test_every_other_shape_is_the_references_by_two_passes(n, order):
    """Round 6's rule for the unflagged order outside the certified shapes (orc_default_order_is_two_pass): the oracle's
    ACORR_CANONICAL mode IS its ACORR_REFERENCE mode there -- R[], coefficients, records, residuals -- no certificate runs
    (the counters stay at zero), and ACORR_CHUNK_TREE (FLACENC_HIP_FLAG_CANONICAL_SUM_ORDER) keeps the 16-sample chunk
    tree, whose R[] differs."""
    rng = np.random.default_rng(n + order)
    x = _corpus(rng, 8, n, 16)
    kw = dict(lpc_order=order)
    orc.cert_stats(reset=True)
    cp, cres, cR, cA = orc.qlpc_batch(x, 16, orc.make_config(acorr=orc.ACORR_CANONICAL, **kw), nthreads=1)
    assert orc.cert_stats() == (0, 0, 0)
    rp, rres, rR, rA = orc.qlpc_batch(x, 16, orc.make_config(acorr=orc.ACORR_REFERENCE, **kw))
    tp, tres, tR, tA = orc.qlpc_batch(x, 16, orc.make_config(acorr=orc.ACORR_CHUNK_TREE, **kw))
    assert np.array_equal(cR.view(np.uint64), rR.view(np.uint64)) and np.array_equal(cA.view(np.uint64), rA.view(np.uint64))
    assert cp.tobytes() == rp.tobytes() and np.array_equal(cres, rres)
    assert not np.array_equal(tR[:, : order + 1].view(np.uint64), rR[:, : order + 1].view(np.uint64))
