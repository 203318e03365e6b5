"""Re-run one seed of tests/test_gpu_parity.py::test_frame_pipeline_config_fuzz and print what differs."""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import test_gpu_parity as T
from flacenc_rs_amd import _capi
from oracle import oracle as orc
h = _capi.Handle(0)
seed = int(sys.argv[1])
rng = np.random.default_rng(9000 + seed)
for trial in range(5):
    n = int(rng.choice([4096, 4096, 4096, 1152, 4608, 256, 2048]))
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
    cfg = _capi.make_frame_config(_capi.make_config(**qcfg), **flags, **fixed)
    got, gres = h.encode_stereo_frames(x, bps, cfg)
    ocfg = orc.make_frame_config(orc.make_config(acorr=orc.ACORR_CANONICAL, **qcfg), **flags,
                                 fixed=orc.make_fixed_config(max_order=fixed["fixed_max_order"], order_sel=fixed["fixed_order_sel"],
                                                             partitions=fixed["fixed_partitions"], sum_mode=orc.SUMABS_CANONICAL))
    want, wres = orc.encode_stereo_frames_cfg(x, bps, ocfg)
    for f in range(x.shape[0]):
        g, w = got[f], want[f]
        diffs = [k for k in ("channel_assignment", "role", "kind", "dc_offset", "bits") if g[k].tolist() != w[k].tolist()]
        for c in range(2):
            for fld in ("order", "shift", "precision", "rice_order", "status", "code_bits", "subframe_bits", "sum_quotients"):
                if int(g["lpc"][c][fld]) != int(w["lpc"][c][fld]):
                    diffs.append((c, fld, int(g["lpc"][c][fld]), int(w["lpc"][c][fld])))
            if g["lpc"][c]["coefs"].tolist() != w["lpc"][c]["coefs"].tolist():
                diffs.append((c, "coefs", g["lpc"][c]["coefs"][:12].tolist(), w["lpc"][c]["coefs"][:12].tolist()))
            if not np.array_equal(gres[f, c], wres[f, c]):
                d = np.flatnonzero(gres[f, c] != wres[f, c])
                diffs.append((c, "residual", len(d), d[:5].tolist()))
        if diffs:
            print("trial", trial, "n", n, "bps", bps, qcfg, flags, fixed)
            print(" frame", f, "got", g["channel_assignment"], g["role"], g["kind"], g["bits"], "want", w["channel_assignment"], w["role"], w["kind"], w["bits"])
            print("  ", diffs)
            # candidate-level view
            params, resid = h.stereo_qlpc_batch(x[f:f + 1], bps, _capi.make_config(**qcfg))
            l, r = x[f]; m, s_ = orc.stereo_to_midside(l, r)
            for role, sig in enumerate([l, r, m, s_]):
                o = orc.estimated_qlpc(sig, bps + (role == 3), orc.make_config(acorr=orc.ACORR_CANONICAL, **qcfg))
                p = params[0, role]
                print("   role", role, "gpu bits", int(p["subframe_bits"]), "orc", o["subframe_bits"], "order", int(p["order"]), o["order"],
                      "shift", int(p["shift"]), o["shift"], "status", int(p["status"]), o["status"], "coefs eq", p["coefs"][:o["order"]].tolist() == o["coefs"].tolist(),
                      "resid eq", np.array_equal(resid[0, role], o["residual"]), "maxabs", int(np.abs(sig).max()))
                if int(p["subframe_bits"]) != o["subframe_bits"]:
                    print("     gpu rice_order", int(p["rice_order"]), "code_bits", int(p["code_bits"]), "params", p["rice_params"][:1 << int(p["rice_order"])].tolist())
                    print("     orc rice_order", o["rice_order"], "code_bits", o["code_bits"], "params", o["rice_params"].tolist())
                    np.save(os.path.join(root, "gpurun_out", f"fuzz_resid_{seed}_{f}_{role}.npy"), o["residual"])
