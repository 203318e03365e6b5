"""The two-pass form (FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER) of the headline launch a few times, for rocprofv3."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from flacenc_rs_amd import _capi
F, n, bps = 24576, 4096, 16
order = int(sys.argv[1]) if len(sys.argv) > 1 else 10
x = torch.from_numpy(_capi.sigen_frames(F, 2, n, bps, 200.0, 0.4, 0.4, seed=0xF1AC0001)).cuda()
results = torch.zeros((F, 752), dtype=torch.uint8, device="cuda")
residual = torch.zeros((F * 2, n), dtype=torch.int32, device="cuda")
h = _capi.Handle(0)
for flags in (_capi.FLAG_REFERENCE_SUM_ORDER, 0):
    cfg = _capi.make_frame_config(_capi.make_config(lpc_order=order, flags=flags), use_fixed=False)
    for _ in range(6):
        h.encode_stereo_frames_device(cfg, x.data_ptr(), F, n, n, bps, results.data_ptr(), residual.data_ptr(), n, stream=0)
    torch.cuda.synchronize()
