"""One BASELINE shape through flacenc_hip_stereo_qlpc_batch_async a few times, for rocprofv3 runs:
    rocprofv3 --kernel-trace --stats -d out -- python3 tools/prof_config.py --n 8192 --order 32 --bps 24 --frames 2048"""
import argparse
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from flacenc_rs_amd import _capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=8192)
ap.add_argument("--order", type=int, default=32)
ap.add_argument("--bps", type=int, default=24)
ap.add_argument("--frames", type=int, default=2048)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--flags", type=int, default=0)
args = ap.parse_args()
h = _capi.Handle(0)
host = _capi.sigen_frames(args.frames, 2, args.n, args.bps, 200.0, 0.4, 0.1, seed=7)
x = torch.from_numpy(host).cuda()
params = torch.empty((args.frames * 4, 352), dtype=torch.uint8, device="cuda")
resid = torch.empty((args.frames * 4, args.n), dtype=torch.int32, device="cuda")
cfg = _capi.make_config(lpc_order=args.order, flags=args.flags)
for _ in range(args.reps):
    h.stereo_qlpc_batch_device(cfg, x.data_ptr(), args.frames, args.n, args.n, args.bps, params.data_ptr(),
                               resid.data_ptr(), args.n, stream=0)
torch.cuda.synchronize()
