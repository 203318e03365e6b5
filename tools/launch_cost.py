"""Kernel time of the headline launch against the batch size: T(F) = fixed + slope * F (HIP events, median of 30)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from flacenc_rs_amd import _capi
n, bps = 4096, 16
Fmax = 24576
x = torch.from_numpy(_capi.sigen_frames(Fmax, 2, n, bps, 200.0, 0.4, 0.4, seed=0xF1AC0001)).cuda()
results = torch.zeros((Fmax, 752), dtype=torch.uint8, device="cuda")
residual = torch.zeros((Fmax * 2, n), dtype=torch.int32, device="cuda")
cfg = _capi.make_frame_config(_capi.make_config(lpc_order=8), use_fixed=False)
h = _capi.Handle(0)
st = torch.cuda.current_stream()
rows = []
for F in (768, 1536, 3072, 6144, 8192, 12288, 24576):
    go = lambda: h.encode_stereo_frames_device(cfg, x.data_ptr(), F, n, n, bps, results.data_ptr(), residual.data_ptr(), n, stream=st.cuda_stream)
    for _ in range(20):
        go()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for a, b in ev:
        a.record(st); go(); b.record(st)
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
    rows.append((F, ms))
    print(f"F={F:6d} rounds {F / 768:6.2f}  {ms * 1e3:8.1f} us  {ms * 1e3 / (F / 768):7.1f} us per round")
A = np.array([[1, f] for f, _ in rows]); y = np.array([m for _, m in rows])
fixed, slope = np.linalg.lstsq(A, y, rcond=None)[0]
print(f"fit: fixed {fixed * 1e3:.1f} us + {slope * 768 * 1e3:.2f} us per round of 768 workgroups")
