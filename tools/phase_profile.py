"""Per-phase latency of the fused QLPC kernel from in-kernel s_memtime stamps (profiling only).

    python tools/phase_profile.py [--frames 8192] [--lpc-order 8] [--block-size 4096]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from flacenc_rs_amd import _capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=8192)
ap.add_argument("--lpc-order", type=int, default=8)
ap.add_argument("--block-size", type=int, default=4096)
ap.add_argument("--bps", type=int, default=16)
ap.add_argument("--signal", default="200,0.4,0.4", help="sine period, sine amplitude, noise amplitude")
ap.add_argument("--flags", type=int, default=0, help="flacenc_hip_qlpc_config.flags (128 = the bare chunk tree)")
ap.add_argument("--use-fixed", action="store_true", help="fixed-LPC candidate on: its selection and "
                "coding pass land in the 'acorr' slot (stamps are rewritten by the last candidate pass)")
args = ap.parse_args()
n, F = args.block_size, args.frames
dev = torch.device("cuda", 0)
sp, sa, na = (float(v) for v in args.signal.split(","))
host = _capi.sigen_frames(F, 2, n, args.bps, sp, sa, na, seed=0xF1AC0001)
x = torch.from_numpy(host).to(dev)
results = torch.empty((F, 752), dtype=torch.uint8, device=dev)
residual = torch.empty((F * 2, n), dtype=torch.int32, device=dev)
stamps = torch.zeros((F * 4, 8), dtype=torch.int64, device=dev)
cfg = _capi.make_frame_config(_capi.make_config(lpc_order=args.lpc_order, flags=args.flags), use_fixed=args.use_fixed)
h = _capi.Handle(0, hooks=True)
for it in range(3):
    h.debug_set_stamps(stamps.data_ptr() if it == 2 else 0)
    h.encode_stereo_frames_device(cfg, x.data_ptr(), F, n, n, args.bps, results.data_ptr(), residual.data_ptr(), n, stream=0)
    torch.cuda.synchronize()
s = stamps.cpu().numpy().astype(np.float64)
d = np.diff(s, axis=1)
names = ["load", "acorr", "levinson+quant", "residual", "(store, 4-candidate mode)", "rice+decide+store", "record"]
tot = s[:, 7] - s[:, 0]
print(f"workgroups {len(s)}; median total {np.median(tot):.0f} cycles; span {s.max() - s.min():.0f} cycles")
for i, nm in enumerate(names):
    print(f"  {nm:16s} median {np.median(d[:, i]):8.0f}  mean {d[:, i].mean():8.0f} cycles  ({100 * d[:, i].mean() / tot.mean():5.1f} %)")
