import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from flacenc_rs_amd import _capi
F, n, bps = 1024, 4096, 16
x = torch.from_numpy(_capi.sigen_frames(F, 2, n, bps, 200.0, 0.4, 0.4, seed=0xF1AC0001)).cuda()
results = torch.zeros((F, 752), dtype=torch.uint8, device="cuda")
residual = torch.zeros((F * 2, n), dtype=torch.int32, device="cuda")
h = _capi.Handle(0, hooks=True)
for flags in (0, 128):
    stats = torch.zeros(3, dtype=torch.int32, device="cuda")
    h.debug_set_cert_stats(stats.data_ptr())
    cfg = _capi.make_frame_config(_capi.make_config(lpc_order=8, flags=flags), use_fixed=False)
    h.encode_stereo_frames_device(cfg, x.data_ptr(), F, n, n, bps, results.data_ptr(), residual.data_ptr(), n, stream=0)
    torch.cuda.synchronize()
    print("flags", flags, "stats", stats.cpu().tolist())
