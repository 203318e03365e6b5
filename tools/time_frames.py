"""ms per call of the frame-level pipeline (flacenc_hip_encode_stereo_frames, then pack) for any shape:
    python tools/time_frames.py --n 8192 --order 24 --bps 24 --frames 2048 [--use-fixed]"""
import argparse, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from flacenc_rs_amd import _capi
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=8192)
ap.add_argument("--order", type=int, default=24)
ap.add_argument("--bps", type=int, default=24)
ap.add_argument("--frames", type=int, default=2048)
ap.add_argument("--use-fixed", action="store_true")
ap.add_argument("--reps", type=int, default=6)
args = ap.parse_args()
F, n = args.frames, args.n
h = _capi.Handle(0)
x = torch.from_numpy(_capi.sigen_frames(F, 2, n, args.bps, 200.0, 0.4, 0.1, seed=7)).cuda()
results = torch.empty((F, 752), dtype=torch.uint8, device="cuda")
resid = torch.empty((F * 2, n), dtype=torch.int32, device="cuda")
stride = (h.frame_bytes_bound(n, args.bps) + 15) // 16 * 16
out = torch.empty((F, stride), dtype=torch.uint8, device="cuda")
lens = torch.empty((F,), dtype=torch.int32, device="cuda")
cfg = _capi.make_frame_config(_capi.make_config(lpc_order=args.order), use_fixed=args.use_fixed)
st = torch.cuda.current_stream()
enc = lambda: h.encode_stereo_frames_device(cfg, x.data_ptr(), F, n, n, args.bps, results.data_ptr(), resid.data_ptr(), n, stream=st.cuda_stream)
pack = lambda: h.pack_stereo_frames_device(x.data_ptr(), F, n, n, results.data_ptr(), resid.data_ptr(), n, args.bps, 96000, 0, 1,
                                           out.data_ptr(), stride, lens.data_ptr(), stream=st.cuda_stream)
for name, fn in (("encode_stereo_frames", enc), ("pack_stereo_frames", pack)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(args.reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); fn(); b.record(st); torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    print(f"n={n} order={args.order} bps={args.bps} use_fixed={args.use_fixed} {name}: median {np.median(ms):.3f} ms -> {F * 2 * n / np.median(ms) / 1e6:.1f} G input samples/s")
