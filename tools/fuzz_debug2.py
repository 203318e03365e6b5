"""Re-create one trial of test_extreme_signals_and_layout_fuzz and compare candidates for a frame."""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import torch
torch.cuda.init()
import test_gpu_parity as T
from flacenc_rs_amd import _capi
from oracle import oracle as orc
h = _capi.Handle(0)
seed, want_trial, frame = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
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
    x = T._extreme_frames(rng, n, bps)
    stride = n + int(rng.choice([0, 0, 4, 8, 3, 5])); rstride = n + int(rng.choice([0, 0, 4, 1]))
    off_in, off_out = int(rng.choice([0, 0, 4, 1, 2])), int(rng.choice([0, 0, 4, 3]))
    first = int(rng.choice([0, 127, 128, 1 << 11, 1 << 16, 1 << 21, 1 << 26, (1 << 31) - x.shape[0]]))
    if trial != want_trial:
        continue
    print(n, bps, qcfg, use_fixed, fx)
    l, r = x[frame]; m, s = orc.stereo_to_midside(l, r)
    params, resid = h.stereo_qlpc_batch(x[frame:frame + 1], bps, _capi.make_config(**qcfg))
    pR = None
    for role, sig in enumerate([l, r, m, s]):
        b = bps + (role == 3)
        o = orc.estimated_qlpc(sig, b, orc.make_config(acorr=orc.ACORR_CANONICAL, **qcfg))
        p = params[0, role]
        print("role", role, "min/max", int(sig.min()), int(sig.max()), "gpu status", int(p["status"]), "bits", int(p["subframe_bits"]),
              "| orc status", o["status"], "bits", o["subframe_bits"], "order", int(p["order"]), o["order"],
              "coefs eq", p["coefs"][:o["order"]].tolist() == o["coefs"].tolist())
        if role == 3:
            pp, rr, R, A = h.qlpc_batch(sig[None], np.array([b], np.uint8), _capi.make_config(**qcfg), want_fp=True)
            print("  gpu R[:4]", R[0, :4], "\n  orc R[:4]", o["autocorr"][:4])
            print("  gpu a[:4]", A[0, :4], "\n  orc a[:4]", o["lpc_coefs"][:4] if "lpc_coefs" in o else None)
            print("  gpu coefs", pp[0]["coefs"][:8].tolist(), "orc coefs", o["coefs"][:8].tolist(), "shift", int(pp[0]["shift"]), o["shift"])
            print("  resid eq", np.array_equal(rr[0], o["residual"]), "gpu resid[24:30]", rr[0][24:30].tolist(), "orc", o["residual"][24:30].tolist())
            print("  gpu rice_order", int(pp[0]["rice_order"]), "code_bits", int(pp[0]["code_bits"]), "sum_q", int(pp[0]["sum_quotients"]), "params", pp[0]["rice_params"][:4].tolist())
            print("  orc rice_order", o["rice_order"], "code_bits", o["code_bits"], "sum_q", o["sum_quotients"], "params", o["rice_params"][:4].tolist())
            print("  max|resid|", int(np.abs(o["residual"].astype(np.int64)).max()))
