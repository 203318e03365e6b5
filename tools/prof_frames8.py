"""8-channel Independent frames pipeline (BASELINE configs[3]) in a loop, for rocprofv3."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flacenc_rs_amd import _capi
h = _capi.Handle(0)
frames, ch, n, bps, order = 2048, 8, 4096, 16, 10
x = torch.from_numpy(_capi.sigen_frames(frames, ch, n, bps, 200.0, 0.4, 0.1, seed=7)).cuda()
res = torch.empty((frames * ch, 368), dtype=torch.uint8, device="cuda")
resid = torch.empty((frames * ch, n), dtype=torch.int32, device="cuda")
stride = int(h._lib.flacenc_hip_frame_bytes_bound(ch, n, bps))
out = torch.empty((frames, stride), dtype=torch.uint8, device="cuda")
lens = torch.zeros(frames, dtype=torch.int32, device="cuda")
cfg = _capi.make_frame_config(_capi.make_config(lpc_order=order), use_fixed=True)
for _ in range(6):
    h._check(h._lib.flacenc_hip_encode_frames_async(h._h, cfg, x.data_ptr(), frames, ch, n, n, bps, res.data_ptr(), resid.data_ptr(), n, None))
    h._check(h._lib.flacenc_hip_pack_frames_async(h._h, x.data_ptr(), frames, ch, n, n, res.data_ptr(), resid.data_ptr(), n, bps, 44100, 0, 1, out.data_ptr(), stride, lens.data_ptr(), None))
torch.cuda.synchronize()
