"""In-kernel stamps of bigblock_acorr_kernel (profiling hook): where a pass spends its cycles.
    python tools/phase_profile_bigblock.py --n 8192 --order 24"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flacenc_rs_amd import _capi
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=8192)
ap.add_argument("--order", type=int, default=24)
ap.add_argument("--frames", type=int, default=2048)
args = ap.parse_args()
h = _capi.Handle(0, hooks=True)
F, n = args.frames, args.n
x = torch.from_numpy(_capi.sigen_frames(F, 2, n, 24, 200.0, 0.4, 0.1, seed=7)).cuda()
params = torch.empty((F * 4, 352), dtype=torch.uint8, device="cuda")
resid = torch.empty((F * 4, n), dtype=torch.int32, device="cuda")
stamps = torch.zeros((F * 4, 8), dtype=torch.int64, device="cuda")
cfg = _capi.make_config(lpc_order=args.order)
for it in range(3):
    h.debug_set_stamps(stamps.data_ptr() if it == 2 else 0)
    h.stereo_qlpc_batch_device(cfg, x.data_ptr(), F, n, n, 24, params.data_ptr(), resid.data_ptr(), n, stream=0)
    torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.float64)
names = ["start->barrier", "load issue+wait", "window staging", "barrier", "compute pass 0", "rest (other passes)"]
d = np.diff(s[:, :6], axis=1)
for i, nm in enumerate(names[:5]):
    print(f"{nm:22s} median {np.median(d[:, i]):9.0f} mean {d[:, i].mean():9.0f} cycles")
print(f"{'whole kernel (wave)':22s} median {np.median(s[:, 5] - s[:, 0]):9.0f}")
for role in range(4):
    print("role", role, "compute pass 0 median", np.median(d[role::4, 3]), " load", np.median(d[role::4, 1]))
