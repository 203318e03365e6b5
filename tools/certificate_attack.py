"""Attack on the order certificate: a hill-climber over signal parameters that
MAXIMISES  actual / bound, where

    actual_i = | a^_ref[i] - a^_kernel[i] |   the unquantised coefficients the floating-point recursion gives on the
                                              reference's sums (lpc.rs:533-548) and on the fused kernel's lane-order sums
                                              -- order effect, second order and both recursions' rounding, all of it;
    bound_i  = the certificate's second-tier bound da[i] (safety 2 included; orc_certificate_bounds).

A ratio above 1 would be a counter-example to what the certificate relies on; the soak corpora of round 5 never exceeded
0.007.  Also measured, with exact rational arithmetic: c_L = |T a^ - r|_inf / (P^2 u R0 (1 + |a^|_1)), the residual constant
of the floating-point recursion that the stated bound ASSUMES (<= 11 would do; flacenc_oracle.c, orc_quant_certified).

    python tools/certificate_attack.py [--seconds 120] [--seed 1] [--orders 8,10,12] [--n 4096]

--n: the block size, 4096 or 4608 (the certified shapes).  The allowance the factor 2 leaves the two recursions is
0.77 F_i, i.e. c_L <= 0.77 (n + 96) / (2 P^2): 11 at (4096, 12).

CPU only (the oracle is the statement of the certificate; tests/test_gpu_certified_order.py holds the GPU to it bit for bit).
"""
import argparse
import os
import sys
import time
from fractions import Fraction

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402

N = 4096


def synth(p, bps, rng_noise):
    """A parametrised subframe: up to three partials (period, amplitude, phase, decay), DC, a noise floor, soft clipping."""
    t = np.arange(N, dtype=np.float64)
    x = np.full(N, p["dc"])
    for k in range(3):
        x = x + p["amp%d" % k] * np.exp(-t * p["decay%d" % k]) * np.sin(2 * np.pi * t / p["per%d" % k] + p["ph%d" % k])
    x = x + rng_noise * 10.0 ** p["noise_db"]
    x = np.clip(x * p["gain"], -p["clip"], p["clip"])
    full = float(1 << (bps - 1))
    return np.clip(np.rint(x * full), -full, full - 1).astype(np.int32)


def random_params(rng):
    p = {"dc": rng.uniform(-0.3, 0.3) * (rng.random() < 0.3), "noise_db": rng.uniform(-6.0, -0.5), "gain": rng.uniform(0.3, 1.5),
         "clip": rng.uniform(0.5, 0.999)}
    for k in range(3):
        p["amp%d" % k] = rng.uniform(0.0, 0.6) * (k == 0 or rng.random() < 0.5)
        p["per%d" % k] = 10.0 ** rng.uniform(0.4, 3.3)
        p["ph%d" % k] = rng.uniform(0, 2 * np.pi)
        p["decay%d" % k] = (10.0 ** rng.uniform(-5, -2.5)) * (rng.random() < 0.3)
    return p


def mutate(p, rng, scale):
    q = dict(p)
    for key in rng.choice(list(q), size=rng.integers(1, 4), replace=False):
        if key.startswith("per"):
            q[key] = float(np.clip(q[key] * np.exp(rng.normal(0, 0.3 * scale)), 2.2, 3000.0))
        elif key.startswith("amp"):
            q[key] = float(np.clip(q[key] + rng.normal(0, 0.15 * scale), 0.0, 0.9))
        elif key.startswith("ph"):
            q[key] = float(q[key] + rng.normal(0, 1.0 * scale))
        elif key.startswith("decay"):
            q[key] = float(np.clip(q[key] * np.exp(rng.normal(0, 1.0 * scale)) if q[key] > 0 else 10.0 ** rng.uniform(-5, -3), 0, 0.01))
        elif key == "noise_db":
            q[key] = float(np.clip(q[key] + rng.normal(0, 0.7 * scale), -7.0, -0.3))
        elif key == "gain":
            q[key] = float(np.clip(q[key] * np.exp(rng.normal(0, 0.2 * scale)), 0.05, 3.0))
        elif key == "clip":
            q[key] = float(np.clip(q[key] + rng.normal(0, 0.1 * scale), 0.2, 0.999))
        elif key == "dc":
            q[key] = float(np.clip(q[key] + rng.normal(0, 0.1 * scale), -0.6, 0.6))
    return q


def evaluate(x, bps, order, precision, window):
    """(ratio, detail): max_i |a_ref - a_kernel|_i / da_i for one subframe, or None where the certificate does not apply."""
    kw = dict(lpc_order=order, quant_precision=precision, window=window)
    b = orc.certificate_bounds(x, orc.make_config(**kw))
    if b is None or not np.all(np.isfinite(b["da"])) or np.any(b["da"] <= 0):
        return None
    # only where the certificate CAN pass: every bound below half a quantisation step (a bound of many steps certifies
    # nothing, and on systems with condition numbers near 1 / u the two computed solutions are unrelated anyway)
    amax = float(np.abs(b["a"]).max())
    if not (amax > 0 and np.isfinite(amax)):
        return None
    shift = int(np.clip((precision - 1) - int(np.ceil(np.log2(amax))), 0, 15))
    if float(b["da"].max()) * 2.0 ** shift >= 0.5:
        return None
    _, _, _, a_ref = orc.qlpc_batch(x[None, :], bps, orc.make_config(acorr=orc.ACORR_REFERENCE, **kw))
    _, _, _, a_ker = orc.qlpc_batch(x[None, :], bps, orc.make_config(acorr=orc.ACORR_CHUNK_TREE, **kw))
    act = np.abs(a_ref[0, :order] - a_ker[0, :order])
    assert np.array_equal(a_ker[0, :order], b["a"]), "orc_certificate_bounds and the chunk-tree mode disagree on a[]"
    i = int(np.argmax(act / b["da"]))
    return float(act[i] / b["da"][i]), {"i": i, "actual": float(act[i]), "bound": float(b["da"][i]), "tier1": b["tier1"]}


def levinson_constant(R, a):
    """c_L of one solve: the exact residual of the computed solution, in units of P^2 u R0 (1 + |a|_1)."""
    P = len(a)
    Rf = [Fraction(float(v)) for v in R]
    af = [Fraction(float(v)) for v in a]
    worst = Fraction(0)
    for i in range(P):
        res = sum(Rf[abs(i - j)] * af[j] for j in range(P)) - Rf[i + 1]
        worst = max(worst, abs(res))
    denom = Fraction(P * P) * Fraction(1, 2 ** 53) * Rf[0] * (1 + sum(abs(v) for v in af))
    return float(worst / denom) if denom != 0 else 0.0


def allowed_constant(n, order):
    """The c_L up to which safety 2 covers second order (1.23) + both recursions' rounding (2 c_L P^2 / (n + 96))."""
    return 0.77 * (n + 96) / (2.0 * order * order)


def attack(seconds, seed, orders, log=print, n=None):
    global N
    if n is not None:
        N = int(n)
    rng = np.random.default_rng(seed)
    t_end = time.time() + seconds
    worst = (0.0, None)
    worst_cl = 0.0
    worst_cl_frac = 0.0  # ... as a fraction of what its (n, order) allows
    evals = 0
    while time.time() < t_end:
        order = int(rng.choice(orders))
        precision = int(rng.choice([15, 15, 15, 12, 8, 5]))
        bps = int(rng.choice([16, 16, 16, 24, 12, 8]))
        window = [("tukey", 0.4), ("tukey", 0.4), ("tukey", 0.1), ("tukey", 1.0), "rectangle"][int(rng.integers(0, 5))]
        noise = rng.uniform(-1, 1, N)
        p = random_params(rng)
        r = evaluate(synth(p, bps, noise), bps, order, precision, window)
        evals += 1
        if r is None:
            continue
        best, bdet = r
        scale, stall = 1.0, 0
        while stall < 40 and time.time() < t_end:  # climb from this start
            q = mutate(p, rng, scale)
            x = synth(q, bps, noise)
            r = evaluate(x, bps, order, precision, window)
            evals += 1
            if r is not None and r[0] > best:
                best, bdet, p, stall = r[0], r[1], q, 0
            else:
                stall += 1
                scale = max(0.05, scale * 0.93)
        x = synth(p, bps, noise)
        b = orc.certificate_bounds(x, orc.make_config(lpc_order=order, quant_precision=precision, window=window))
        if b is not None:
            cl = levinson_constant(b["R"], b["a"])
            worst_cl = max(worst_cl, cl)
            worst_cl_frac = max(worst_cl_frac, cl / allowed_constant(N, order))
        if best > worst[0]:
            worst = (best, dict(bdet, order=order, precision=precision, bps=bps, window=window, params=p, signal=x))
            log("  new worst actual/bound %.4f  (order %d, precision %d, %d bit, %s; coefficient %d: |da| %.3e against %.3e)" % (
                best, order, precision, bps, window, bdet["i"], bdet["actual"], bdet["bound"]))
    attack.last_constant_fraction = worst_cl_frac
    return worst, worst_cl, evals


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--orders", default="4,8,10,12")
    ap.add_argument("--save", default=None, help="write the worst case's samples (int32 .npy) here")
    ap.add_argument("--n", type=int, default=4096, help="block size: 4096 or 4608")
    args = ap.parse_args()
    orders = [int(v) for v in args.orders.split(",")]
    worst, cl, evals = attack(args.seconds, args.seed, orders, n=args.n)
    print("n = %d: %d evaluations in %.0f s: worst actual / bound = %.4f; largest recursion constant c_L = %.3f (%.2f of what "
          "its shape allows; the smallest allowance here is %.2f)" % (
              args.n, evals, args.seconds, worst[0], cl, attack.last_constant_fraction, min(allowed_constant(args.n, o) for o in orders)))
    if worst[1]:
        sig = worst[1].pop("signal")
        if args.save:
            np.save(args.save, sig)
        print("worst case:", worst[1])
