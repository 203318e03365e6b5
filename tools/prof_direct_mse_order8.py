import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from flacenc_rs_amd import _capi
h = _capi.Handle(0)
F,n,bps,order=3072,4096,16,8
x = torch.from_numpy(_capi.sigen_frames(F, 2, n, bps, 200.0, 0.4, 0.1, seed=7)).cuda()
params = torch.empty((F * 4, 352), dtype=torch.uint8, device="cuda")
resid = torch.empty((F * 4, n), dtype=torch.int32, device="cuda")
cfg = _capi.make_config(lpc_order=order, use_direct_mse=True, window="rectangle")
for _ in range(8):
    h.stereo_qlpc_batch_device(cfg, x.data_ptr(), F, n, n, bps, params.data_ptr(), resid.data_ptr(), n, stream=0)
torch.cuda.synchronize()
