"""The headline launch (encode_stereo_frames, block 4096, order 8) a few times with no result checks:
for rocprofv3 counter runs of diagnostic builds (FLACENC_HIP_LIB=...)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from flacenc_rs_amd import _capi
F, n, bps = 8192, 4096, 16
x = torch.from_numpy(_capi.sigen_frames(F, 2, n, bps, 200.0, 0.4, 0.4, seed=0xF1AC0001)).cuda()
results = torch.zeros((F, 752), dtype=torch.uint8, device="cuda")
residual = torch.zeros((F * 2, n), dtype=torch.int32, device="cuda")
cfg = _capi.make_frame_config(_capi.make_config(lpc_order=8, flags=int(os.environ.get("FLACENC_FLAGS", "0"))), use_fixed=False)
h = _capi.Handle(0)
for _ in range(4):
    h.encode_stereo_frames_device(cfg, x.data_ptr(), F, n, n, bps, results.data_ptr(), residual.data_ptr(), n, stream=0)
torch.cuda.synchronize()
