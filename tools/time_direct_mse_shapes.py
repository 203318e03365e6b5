import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from flacenc_rs_amd import _capi
h = _capi.Handle(0)
def run(F, n, bps, order, steps):
    x = torch.from_numpy(_capi.sigen_frames(F, 2, n, bps, 200.0, 0.4, 0.1, seed=7)).cuda()
    params = torch.empty((F * 4, 352), dtype=torch.uint8, device="cuda"); resid = torch.empty((F * 4, n), dtype=torch.int32, device="cuda")
    cfg = _capi.make_config(lpc_order=order, use_direct_mse=True, mae_optimization_steps=steps, window="rectangle")
    st = torch.cuda.current_stream()
    go = lambda: h.stereo_qlpc_batch_device(cfg, x.data_ptr(), F, n, n, bps, params.data_ptr(), resid.data_ptr(), n, stream=st.cuda_stream)
    go(); torch.cuda.synchronize(); ms = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); go(); b.record(st); torch.cuda.synchronize(); ms.append(a.elapsed_time(b))
    print(f"n={n} order={order} irls={steps} F={F}: {np.median(ms):.3f} ms  ({np.median(ms)*1e3/(F*4)*2048:.1f} us per subframe-slot at 2048 concurrent)", flush=True)
for n in (1024, 4096):
    for order in (4, 8, 11, 12):
        run(3072, n, 16, order, 0)
run(12288, 4096, 16, 8, 0)
