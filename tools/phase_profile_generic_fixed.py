"""Phase stamps of the generic kernel's fixed-LPC launch inside flacenc_hip_encode_stereo_frames (the last generic
launch of the call writes the stamps last)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from flacenc_rs_amd import _capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=16384)
ap.add_argument("--lpc-order", type=int, default=10)
ap.add_argument("--block-size", type=int, default=1152)
ap.add_argument("--bps", type=int, default=16)
args = ap.parse_args()
n, F = args.block_size, args.frames
x = torch.from_numpy(_capi.sigen_frames(F, 2, n, args.bps, 200.0, 0.4, 0.1, seed=7)).cuda()
results = torch.empty((F, 752), dtype=torch.uint8, device="cuda")
resid = torch.empty((F * 2, n), dtype=torch.int32, device="cuda")
stamps = torch.zeros((F * 4, 8), dtype=torch.int64, device="cuda")
cfg = _capi.make_frame_config(_capi.make_config(lpc_order=args.lpc_order), use_fixed=True)
h = _capi.Handle(0, hooks=True)
for it in range(3):
    h.debug_set_stamps(stamps.data_ptr() if it == 2 else 0)
    h.encode_stereo_frames_device(cfg, x.data_ptr(), F, n, n, args.bps, results.data_ptr(), resid.data_ptr(), n, stream=0)
    torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.float64)
d = np.diff(s, axis=1)
names = ["load", "selector sums + choice", "(none)", "residual", "rice tables", "rice levels", "store+record"]
tot = s[:, 7] - s[:, 0]
print(f"workgroups {len(s)}; median total {np.median(tot):.0f} cycles")
for i, nm in enumerate(names):
    print(f"  {nm:24s} median {np.median(d[:, i]):8.0f}  ({100 * d[:, i].mean() / tot.mean():5.1f} %)")
