"""ms per call (HIP events) of one stereo QLPC batch shape: FLACENC_HIP_LIB=... python tools/time_config.py --n 8192 --order 24"""
import argparse, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from flacenc_rs_amd import _capi
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=8192)
ap.add_argument("--order", type=int, default=24)
ap.add_argument("--bps", type=int, default=24)
ap.add_argument("--frames", type=int, default=0)
ap.add_argument("--reps", type=int, default=12)
ap.add_argument("--sum-order", choices=["canonical", "reference", "nightly"], default="canonical")
args = ap.parse_args()
F = args.frames or (16777216 // args.n)
h = _capi.Handle(0)
x = torch.from_numpy(_capi.sigen_frames(F, 2, args.n, args.bps, 200.0, 0.4, 0.1, seed=7)).cuda()
params = torch.empty((F * 4, 352), dtype=torch.uint8, device="cuda")
resid = torch.empty((F * 4, args.n), dtype=torch.int32, device="cuda")
cfg = _capi.make_config(lpc_order=args.order, flags={"canonical": 0, "reference": _capi.FLAG_REFERENCE_SUM_ORDER,
                                                     "nightly": _capi.FLAG_NIGHTLY_SUM_ORDER}[args.sum_order])
st = torch.cuda.current_stream()
go = lambda: h.stereo_qlpc_batch_device(cfg, x.data_ptr(), F, args.n, args.n, args.bps, params.data_ptr(), resid.data_ptr(), args.n, stream=st.cuda_stream)
for _ in range(3):
    go()
torch.cuda.synchronize()
ms = []
for _ in range(args.reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st); go(); b.record(st); torch.cuda.synchronize()
    ms.append(a.elapsed_time(b))
print(f"{os.path.basename(os.environ.get('FLACENC_HIP_LIB', 'default')):28s} n={args.n} order={args.order} {args.sum_order}: median {np.median(ms):.4f} ms  min {min(ms):.4f}  -> {F * 2 * args.n / np.median(ms) / 1e6:.1f} G input samples/s")
