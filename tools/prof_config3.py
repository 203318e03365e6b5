import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from flacenc_rs_amd import _capi
h = _capi.Handle(0)
frames, n, bps, order = 2048, 8192, 24, 32
host = _capi.sigen_frames(frames, 2, n, bps, 200.0, 0.4, 0.1, seed=7)
x = torch.from_numpy(host).cuda()
params = torch.empty((frames * 4, 352), dtype=torch.uint8, device="cuda")
resid = torch.empty((frames * 4, n), dtype=torch.int32, device="cuda")
cfg = _capi.make_config(lpc_order=order)
for _ in range(6):
    h.stereo_qlpc_batch_device(cfg, x.data_ptr(), frames, n, n, bps, params.data_ptr(), resid.data_ptr(), n, stream=0)
torch.cuda.synchronize()
