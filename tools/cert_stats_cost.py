import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from flacenc_rs_amd import _capi
F, n, bps = 98304, 4096, 16
x = torch.from_numpy(_capi.sigen_frames(F, 2, n, bps, 200.0, 0.4, 0.4, seed=0xF1AC0001)).cuda()
results = torch.zeros((F, _capi.FRAME_RESULT_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
residual = torch.zeros((F * 2, n), dtype=torch.int32, device="cuda")
h = _capi.Handle(0, hooks=True)
cfg = _capi.make_frame_config(_capi.make_config(lpc_order=8), use_fixed=False)
st = torch.zeros(3, dtype=torch.int32, device="cuda")
go = lambda: h.encode_stereo_frames_device(cfg, x.data_ptr(), F, n, n, bps, results.data_ptr(), residual.data_ptr(), n, stream=0)
def timeit():
    for _ in range(5): go()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); go(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))
for rep in range(3):
    h.debug_set_cert_stats(0); t0 = timeit()
    h.debug_set_cert_stats(st.data_ptr()); t1 = timeit()
    print("without counters %.4f ms, with %.4f ms" % (t0, t1))
