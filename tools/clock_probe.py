"""Effective shader clock and resident-wave concurrency of the fused kernel: in-kernel s_memtime stamps
(shader cycles) against the HIP-event duration of the same launch.  Diagnostic only.
    FLACENC_HIP_LIB=... python tools/clock_probe.py [--use-fixed]"""
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
ap.add_argument("--use-fixed", action="store_true")
args = ap.parse_args()
n, F, bps = 4096, args.frames, 16
dev = torch.device("cuda", 0)
x = torch.from_numpy(_capi.sigen_frames(F, 2, n, bps, 200.0, 0.4, 0.4, seed=0xF1AC0001)).to(dev)
results = torch.empty((F, 752), dtype=torch.uint8, device=dev)
residual = torch.empty((F * 2, n), dtype=torch.int32, device=dev)
stamps = torch.zeros((F * 4, 8), dtype=torch.int64, device=dev)
cfg = _capi.make_frame_config(_capi.make_config(lpc_order=args.lpc_order), use_fixed=args.use_fixed)
h = _capi.Handle(0, hooks=True)
st = torch.cuda.current_stream()


def launch():
    h.encode_stereo_frames_device(cfg, x.data_ptr(), F, n, n, bps, results.data_ptr(), residual.data_ptr(), n,
                                  stream=st.cuda_stream)


for _ in range(20):   # warm the clocks
    launch()
torch.cuda.synchronize()
h.debug_set_stamps(stamps.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
launch()
e1.record(st)
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
s = stamps.cpu().numpy().astype(np.float64)
life = s[:, 7] - s[:, 0]
# s_memtime counters are per XCD and not aligned with each other; workgroup b runs on XCD b % 8
xcd = (np.arange(len(s)) // 4) % 8
spans = np.array([s[xcd == k, 7].max() - s[xcd == k, 0].min() for k in range(8)])
span = float(np.median(spans))
print(f"lib {os.environ.get('FLACENC_HIP_LIB', 'default')}")
print(f"kernel {ms:.4f} ms (with stamps); stamp span {span:.0f} cycles -> effective clock {span / ms / 1e6:.3f} GHz")
print(f"per-XCD spans {spans.astype(int).tolist()}")
print(f"wave lifetime median {np.median(life):.0f} mean {life.mean():.0f} cycles; resident waves (sum of lifetimes / span) "
      f"{life.sum() / span:.0f} = {life.sum() / span / 256:.2f} per CU")
d = np.diff(s, axis=1)
names = ["load", "acorr", "levinson", "residual", "-", "rice+decide+store", "record"]
print("  " + "  ".join(f"{nm} {100 * d[:, i].mean() / life.mean():.1f}%" for i, nm in enumerate(names)))
