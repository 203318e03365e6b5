"""CPU soak of the order certificate through the oracle (the test tests/test_certificate_cpu.py at scale):
    python tools/certificate_soak.py [rounds=400] [seed=77] [sizes=4096,4096,4096,4608]
(sizes: the block sizes the rounds cycle through, of the certified shapes 4096 / 4608)
prints subframes / tier-2 / recomputed / certified-but-different (must be 0) / bare-order-differs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle as orc
import test_certificate_cpu as T

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 77)
sizes = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [4608, 4096, 4096, 4096]
total = tier2 = redone = bad = tree = 0
for rnd in range(rounds):
    n = sizes[rnd % len(sizes)]
    order = int(rng.integers(1, 13)); precision = int(rng.integers(3, 16)); bps = int(rng.choice([8, 12, 16, 16, 16, 20, 24]))
    window = [("tukey", 0.4), ("tukey", 0.1), ("tukey", 1.0), "rectangle"][rnd % 4]
    x = T._corpus(rng, 160, n, bps)
    kw = dict(lpc_order=order, quant_precision=precision, window=window)
    orc.cert_stats(reset=True)
    cp, cres, _, _ = orc.qlpc_batch(x, bps, orc.make_config(acorr=orc.ACORR_CANONICAL, **kw), nthreads=1, want_fp=False)
    st = orc.cert_stats()
    rp, rres, _, _ = orc.qlpc_batch(x, bps, orc.make_config(acorr=orc.ACORR_REFERENCE, **kw), want_fp=False)
    tp, _, _, _ = orc.qlpc_batch(x, bps, orc.make_config(acorr=orc.ACORR_CHUNK_TREE, **kw), want_fp=False)
    bad += int(((cp["coefs"] != rp["coefs"]).any(axis=1) | (cp["shift"] != rp["shift"]) | (cp["order"] != rp["order"])).sum())
    bad += int((cres != rres).any(axis=1).sum())
    tree += int(((tp["coefs"] != rp["coefs"]).any(axis=1) | (tp["shift"] != rp["shift"])).sum())
    total += st[0]; tier2 += st[1]; redone += st[2]
print(f"{total} subframes: {tier2} needed the rows of T^-1, {redone} recomputed ({redone / total:.4f}), "
      f"{bad} certified-but-different, bare kernel order differs in {tree} ({tree / total:.4f})")
