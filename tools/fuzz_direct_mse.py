"""Seed sweep of the experimental estimators (use_direct_mse / mae_optimization_steps) against the
oracle: random block sizes, orders, precisions, windows, bits per sample, IRLS steps and material through
flacenc_hip_qlpc_batch -- R[], the unquantised solution, records and residuals bit for bit, decode round trip.

    python tools/fuzz_direct_mse.py [first_seed] [last_seed]
"""
import os, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import torch
torch.cuda.init()
from flacenc_rs_amd import _capi
from oracle import oracle as orc

h = _capi.Handle(0)
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad = 0
checked = irls = degenerate = 0
t0 = time.time()
for seed in range(lo, hi):
    rng = np.random.default_rng(424200 + seed)
    n = int(rng.choice([4096, 4096, 8192, 16384, 4608, 1152, 576, 256, 100, 64, 2048, 1000, 20000]))
    bps = int(rng.choice([8, 12, 16, 16, 20, 24]))
    order = int(rng.choice([1, 2, 4, 8, 8, 10, 12, 16, 24, 32]))
    order = min(order, 32, n - 1)
    steps = int(rng.choice([0, 0, 0, 1, 2, 3]))  # (blocks above 16384 samples: the IRLS weights live in HBM scratch, round 4)
    qcfg = dict(lpc_order=order, quant_precision=int(rng.integers(4, 16)),
                window=("rectangle" if rng.random() < 0.6 else ("tukey", float(np.round(rng.random(), 2)))),
                max_rice_parameter=int(rng.choice([7, 14, 15, 30, 30])))
    ns = int(rng.integers(1, 6))
    amp = float(rng.choice([0.0, 0.003, 0.2, 0.6, 0.9]))
    namp = float(min(0.99 - amp, rng.choice([0.0, 0.002, 0.1, 0.5])))
    x = _capi.sigen_frames(ns, 1, n, bps, float(rng.uniform(2.2, 400.0)), amp, namp, seed=int(rng.integers(1, 1 << 30)))
    x = np.ascontiguousarray(x.reshape(ns, n))
    if rng.random() < 0.15:
        x[0] = int(rng.integers(-50, 50))          # constant block: the regulariser path
    tag = (seed, n, bps, qcfg, steps, ns, amp, namp)
    try:
        gp, gres, gR, gA = h.qlpc_batch(x, bps, _capi.make_config(use_direct_mse=True, mae_optimization_steps=steps, **qcfg),
                                        want_fp=True)
        rp, rres, rR, rA = orc.qlpc_batch(x, bps, orc.make_config(use_direct_mse=True, mae_optimization_steps=steps, **qcfg))
        assert np.array_equal(gp["status"], rp["status"]), "status"
        ok = rp["status"] == 0
        checked += int(ok.sum()); irls += int(steps > 0) * int(ok.sum()); degenerate += int((~ok).sum())
        assert np.array_equal(gR.view(np.uint64)[ok], rR.view(np.uint64)[ok]), "R[] bits"
        assert np.array_equal(gA.view(np.uint64)[ok], rA.view(np.uint64)[ok]), "solution bits"
        for f in ("order", "shift", "precision", "rice_order", "code_bits", "subframe_bits", "sum_quotients"):
            assert np.array_equal(gp[f][ok], rp[f][ok]), f
        assert np.array_equal(gp["coefs"][ok], rp["coefs"][ok]), "coefs"
        assert np.array_equal(gp["rice_params"][ok], rp["rice_params"][ok]), "rice_params"
        assert np.array_equal(gres[ok], rres[ok]), "residual"
        for k in np.nonzero(ok)[0]:
            o = int(gp["order"][k])
            assert np.array_equal(orc.decode_lpc(x[k][:o], gp["coefs"][k][:o], int(gp["shift"][k]), gres[k]), x[k]), "round trip"
    except Exception as e:  # noqa: BLE001
        bad += 1
        print("FAIL", tag, str(e)[:300], flush=True)
        if bad > 5:
            break
print("done", seed, "failures", bad, "in", round(time.time() - t0), "s;", checked, "subframes compared bit for bit,", irls,
      "of them with IRLS steps,", degenerate, "with a non-zero status on both sides")
