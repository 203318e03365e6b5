"""Throughput of the whole on-device frame pipeline: encode_stereo_frames (+ fixed-LPC candidate)
followed by pack_stereo_frames (Frame::write), inputs and outputs resident in HBM.

    python tools/bench_pack.py [--frames 8192] [--use-fixed] [--steps 20]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from flacenc_rs_amd import _capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=8192)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--lpc-order", type=int, default=8)
ap.add_argument("--bps", type=int, default=16)
ap.add_argument("--use-fixed", action="store_true")
args = ap.parse_args()
F, n, bps = args.frames, 4096, args.bps
host = _capi.sigen_frames(F, 2, n, bps, 200.0, 0.4, 0.4, seed=0xF1AC0001)
x = torch.from_numpy(host).cuda()
res = torch.empty((F, 752), dtype=torch.uint8, device="cuda")
resid = torch.empty((F * 2, n), dtype=torch.int32, device="cuda")
h = _capi.Handle(0)
stride = h.frame_bytes_bound(n, bps)
out = torch.empty((F, stride), dtype=torch.uint8, device="cuda")
lens = torch.zeros(F, dtype=torch.int32, device="cuda")
cfg = _capi.make_frame_config(_capi.make_config(lpc_order=args.lpc_order), use_fixed=args.use_fixed)
st = torch.cuda.current_stream()


def enc():
    h.encode_stereo_frames_device(cfg, x.data_ptr(), F, n, n, bps, res.data_ptr(), resid.data_ptr(), n,
                                  stream=st.cuda_stream)


def pack():
    h.pack_stereo_frames_device(x.data_ptr(), F, n, n, res.data_ptr(), resid.data_ptr(), n, bps, 44100, 0, 1,
                                out.data_ptr(), stride, lens.data_ptr(), stream=st.cuda_stream)


def fused():
    h.encode_pack_stereo_frames_device(cfg, x.data_ptr(), F, n, n, bps, 44100, 0, 1, res.data_ptr(), out.data_ptr(),
                                       stride, lens.data_ptr(), stream=st.cuda_stream)


for _ in range(3):
    enc()
    pack()
    fused()
torch.cuda.synchronize()
evf = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
for k in range(args.steps):
    evf[k].record(st)
    fused()
evf[args.steps].record(st)
torch.cuda.synchronize()
t_fused = evf[0].elapsed_time(evf[args.steps]) / args.steps
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * args.steps)]
for k in range(args.steps):
    ev[3 * k].record(st)
    enc()
    ev[3 * k + 1].record(st)
    pack()
    ev[3 * k + 2].record(st)
torch.cuda.synchronize()
t_enc = float(np.mean([ev[3 * k].elapsed_time(ev[3 * k + 1]) for k in range(args.steps)]))
t_pack = float(np.mean([ev[3 * k + 1].elapsed_time(ev[3 * k + 2]) for k in range(args.steps)]))
total_bytes = int(lens.sum().item())
samples = F * 2 * n
print(json.dumps({
    "frames": F, "use_fixed": args.use_fixed,
    "encode_ms": round(t_enc, 4), "pack_ms": round(t_pack, 4), "encode_pack_one_call_ms": round(t_fused, 4),
    "encode_pack_one_call_Msamples_per_s": round(samples / (t_fused * 1e-3) / 1e6, 1),
    "pipeline_Msamples_per_s": round(samples / ((t_enc + t_pack) * 1e-3) / 1e6, 1),
    "pack_Msamples_per_s": round(samples / (t_pack * 1e-3) / 1e6, 1),
    "flac_bytes": total_bytes, "compression_ratio": round(total_bytes / (samples * bps / 8), 4),
    # pack kernel: reads 2 residual rows + (a little of) the input, writes the frame bytes
    "pack_hbm_GBps": round((samples * 4 + total_bytes) / (t_pack * 1e-3) / 1e9, 1),
}))
