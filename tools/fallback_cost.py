"""Cost of the certificate's in-kernel fallback: near-pure tones (every subframe recomputed from the reference's chains)
against the same launch with the certificate off.  gpurun -- python tools/fallback_cost.py"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import util
from flacenc_rs_amd import _capi
F, n, bps = 24576, 4096, 16
base = np.stack([util.sine_noise(n, bps, 57.3, 0.6, 5e-4, 400 + i, phase=0.1 * i) for i in range(64)])
rng = np.random.default_rng(1)
pick = rng.integers(0, 64, size=(F, 2))
x = torch.from_numpy(np.ascontiguousarray(np.stack([base[pick[:, 0]], base[pick[:, 1]]], axis=1))).cuda()
results = torch.zeros((F, 752), dtype=torch.uint8, device="cuda"); residual = torch.zeros((F * 2, n), dtype=torch.int32, device="cuda")
h = _capi.Handle(0, hooks=True); h.debug_set_adaptive_order(False)
for order in (8, 12):
    for flags in (0, _capi.FLAG_CANONICAL_SUM_ORDER):
        cfg = _capi.make_frame_config(_capi.make_config(lpc_order=order, flags=flags), use_fixed=False)
        go = lambda: h.encode_stereo_frames_device(cfg, x.data_ptr(), F, n, n, bps, results.data_ptr(), residual.data_ptr(), n, stream=0)
        for _ in range(4): go()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]
        for a, b in ev:
            a.record(); go(); b.record()
        torch.cuda.synchronize()
        st = torch.zeros(3, dtype=torch.int32, device="cuda")
        h.debug_set_cert_stats(st.data_ptr()); go(); torch.cuda.synchronize(); h.debug_set_cert_stats(0)
        print(os.path.basename(os.environ.get("FLACENC_HIP_LIB", "default")), "order", order, "flags", flags, "%.3f ms" % float(np.median([a.elapsed_time(b) for a, b in ev])), st.cpu().tolist())
