"""ms per launch (HIP events) of the frame packer alone on the bench workload:
FLACENC_HIP_LIB=... python tools/time_pack.py [--frames 24576] [--n 4096] [--order 8] [--bps 16] [--use-fixed]"""
import argparse, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from flacenc_rs_amd import _capi
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=24576)
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--order", type=int, default=8)
ap.add_argument("--bps", type=int, default=16)
ap.add_argument("--reps", type=int, default=12)
ap.add_argument("--use-fixed", action="store_true")
ap.add_argument("--signal", default="200,0.4,0.4")
args = ap.parse_args()
F, n = args.frames, args.n
sp, sa, na = (float(v) for v in args.signal.split(","))
h = _capi.Handle(0)
x = torch.from_numpy(_capi.sigen_frames(F, 2, n, args.bps, sp, sa, na, seed=0xF1AC0001)).cuda()
results = torch.empty((F, 752), dtype=torch.uint8, device="cuda")
resid = torch.empty((F * 2, n), dtype=torch.int32, device="cuda")
cfg = _capi.make_frame_config(_capi.make_config(lpc_order=args.order), use_fixed=args.use_fixed)
stride = (h.frame_bytes_bound(n, args.bps) + 15) // 16 * 16
out = torch.empty((F, stride), dtype=torch.uint8, device="cuda")
lens = torch.empty((F,), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream()
h.encode_stereo_frames_device(cfg, x.data_ptr(), F, n, n, args.bps, results.data_ptr(), resid.data_ptr(), n, stream=st.cuda_stream)
go = lambda: h.pack_stereo_frames_device(x.data_ptr(), F, n, n, results.data_ptr(), resid.data_ptr(), n, args.bps, 44100, 0, 1,
                                         out.data_ptr(), stride, lens.data_ptr(), stream=st.cuda_stream)
t_end = torch.cuda.Event(enable_timing=True)
for _ in range(40):  # clock spin-up
    go()
torch.cuda.synchronize()
ms = []
for _ in range(args.reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st); go(); b.record(st); torch.cuda.synchronize()
    ms.append(a.elapsed_time(b))
nbytes = int(lens.sum().item())
import zlib
print(f"{os.path.basename(os.environ.get('FLACENC_HIP_LIB', 'default')):28s} pack F={F} n={n}: median {np.median(ms):.4f} ms  min {min(ms):.4f}"
      f"  -> {F * 2 * n / np.median(ms) / 1e6:.1f} G samples/s; bytes {nbytes} crc32 {zlib.crc32(out.cpu().numpy()[:64].tobytes()) & 0xffffffff:08x}")
