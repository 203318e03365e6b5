"""In-kernel stamps of bigblock_residual_kernel (profiling hook): phases of a wave and the dispatch timeline.
    python tools/phase_profile_bigres.py --n 8192 --order 24 --frames 6144"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flacenc_rs_amd import _capi
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=8192)
ap.add_argument("--order", type=int, default=24)
ap.add_argument("--bps", type=int, default=24)
ap.add_argument("--frames", type=int, default=6144)
args = ap.parse_args()
h = _capi.Handle(0, hooks=True)
F, n = args.frames, args.n
x = torch.from_numpy(_capi.sigen_frames(F, 2, n, args.bps, 200.0, 0.4, 0.1, seed=7)).cuda()
params = torch.empty((F * 4, 352), dtype=torch.uint8, device="cuda")
resid = torch.empty((F * 4, n), dtype=torch.int32, device="cuda")
stamps = torch.zeros((F * 4, 8), dtype=torch.int64, device="cuda")
cfg = _capi.make_config(lpc_order=args.order)
for it in range(4):
    h.debug_set_stamps(stamps.data_ptr() if it == 3 else 0)
    h.stereo_qlpc_batch_device(cfg, x.data_ptr(), F, n, n, args.bps, params.data_ptr(), resid.data_ptr(), n, stream=0)
    torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.float64)
names = ["entry -> A operand + seeds ... first split + barrier", "first pass's tiles", "remaining passes", "Rice search"]
d = np.diff(s[:, 1:6], axis=1)
for i, nm in enumerate(names):
    print(f"{nm:56s} median {np.median(d[:, i]):9.0f} mean {d[:, i].mean():9.0f} shader cycles")
print(f"{'whole kernel (wave), shader cycles':56s} median {np.median(s[:, 5] - s[:, 1]):9.0f}")
t0, t1 = s[:, 0], s[:, 7]
print("raw stamps of wave 0:", stamps[0].cpu().numpy().tolist())
print(f"wall clock (100 MHz ticks): launch spans {(t1.max() - t0.min()) / 100:.1f} us; wave lifetime median {np.median(t1 - t0) / 100:.2f} us")
for role in range(4):
    print("role", role, "tiles of pass 0 median", np.median(d[role::4, 1]), " lifetime us", np.median((t1 - t0)[role::4]) / 100)
# occupancy over time: resident workgroups (wave 0 of each) sampled on a grid
w0s, w0e = t0[0::4], t1[0::4]
grid = np.linspace(t0.min(), t1.max(), 41)
res = [(int(((w0s <= g) & (w0e > g)).sum())) for g in grid]
print("resident workgroups over the launch (41 samples):", res)
