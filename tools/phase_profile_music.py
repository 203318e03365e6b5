"""Where a music frame's time goes in the fused kernel: in-kernel stamps (tools/phase_profile.py) on the stereo frames cut
from the reference's real-audio fixtures, with the certificate on (flags 0, order mode pinned to the certified kernel) and off
(FLACENC_HIP_FLAG_CANONICAL_SUM_ORDER).  Phase 2 (stamp 2 -> 3: R[] handed to wave 0, recursion, certificate, its second
tier, the reference's chains, the barrier) is what differs; its distribution shows the three kinds of frame.
    gpurun -- python tools/phase_profile_music.py [order=10] [frames=8192]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from flacenc_rs_amd import _capi

order = int(sys.argv[1]) if len(sys.argv) > 1 else 10
F = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
n, bps = 4096, 16
gold = os.path.join(ROOT, "tests", "golden")
cut = []
for nm in ("sus109", "sus6", "ras22", "ras103"):
    ch = [np.fromfile(os.path.join(gold, "testsignal.%s.ch%d.bin" % (nm, c)), dtype="<i2").astype(np.int32) for c in (0, 1)]
    for t0 in range(0, 8192 - n + 1, 64):
        cut.append(np.stack([ch[0][t0:t0 + n], ch[1][t0:t0 + n]]))
cut = np.stack(cut)
x = torch.from_numpy(np.ascontiguousarray(np.tile(cut, ((F + len(cut) - 1) // len(cut), 1, 1))[:F])).cuda()
results = torch.zeros((F, _capi.FRAME_RESULT_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
residual = torch.zeros((F * 2, n), dtype=torch.int32, device="cuda")
stamps = torch.zeros((F * 4, 8), dtype=torch.int64, device="cuda")
h = _capi.Handle(0, hooks=True)
h.debug_set_adaptive_order(False)
for flags in (0, _capi.FLAG_CANONICAL_SUM_ORDER):
    cfg = _capi.make_frame_config(_capi.make_config(lpc_order=order, flags=flags), use_fixed=False)
    for it in range(3):
        h.debug_set_stamps(stamps.data_ptr() if it == 2 else 0)
        h.encode_stereo_frames_device(cfg, x.data_ptr(), F, n, n, bps, results.data_ptr(), residual.data_ptr(), n, stream=0)
        torch.cuda.synchronize()
    h.debug_set_stamps(0)
    s = stamps.cpu().numpy().astype(np.float64).reshape(F, 4, 8)
    d = np.diff(s, axis=2)
    tot = s[:, :, 7] - s[:, :, 0]
    p2 = d[:, 0, 2]  # wave 0's phase 2
    print("order %d flags %3d: wave lifetime median %.0f mean %.0f; span %.0f" % (order, flags, np.median(tot), tot.mean(), s.max() - s[s > 0].min()))
    names = ["load", "acorr", "levinson+cert", "residual", "-", "rice+decide+store", "record"]
    for i, nm in enumerate(names):
        print("   %-18s mean by role L %7.0f R %7.0f M %7.0f S %7.0f" % (nm, *[d[:, r, i].mean() for r in range(4)]))
    q = np.percentile(p2, [5, 25, 50, 75, 95, 99])
    print("   wave 0 phase 2 percentiles 5/25/50/75/95/99:", " ".join("%.0f" % v for v in q))
    hist, edges = np.histogram(p2, bins=[0, 3000, 6000, 10000, 15000, 25000, 40000, 60000, 90000, 150000, 1e9])
    print("   histogram:", ", ".join("<%s: %d" % ("%.0f" % e if e < 1e9 else "inf", c) for e, c in zip(edges[1:], hist)))
