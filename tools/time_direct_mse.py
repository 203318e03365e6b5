"""ms per call of the experimental estimators (use_direct_mse, IRLS steps) through flacenc_hip_stereo_qlpc_batch:
    python tools/time_direct_mse.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from flacenc_rs_amd import _capi
h = _capi.Handle(0)


def run(F, n, bps, order, steps, window):
    x = torch.from_numpy(_capi.sigen_frames(F, 2, n, bps, 200.0, 0.4, 0.1, seed=7)).cuda()
    params = torch.empty((F * 4, 352), dtype=torch.uint8, device="cuda")
    resid = torch.empty((F * 4, n), dtype=torch.int32, device="cuda")
    cfg = _capi.make_config(lpc_order=order, use_direct_mse=True, mae_optimization_steps=steps, window=window,
                            flags=_capi.FLAG_ALLOW_ORDER_32 if order > 24 else 0)
    st = torch.cuda.current_stream()
    go = lambda: h.stereo_qlpc_batch_device(cfg, x.data_ptr(), F, n, n, bps, params.data_ptr(), resid.data_ptr(), n,
                                            stream=st.cuda_stream)
    go(); torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); go(); b.record(st); torch.cuda.synchronize(); ms.append(a.elapsed_time(b))
    print(f"direct_mse n={n} bps={bps} order={order} irls={steps} {window}: {np.median(ms):.3f} ms -> "
          f"{F * 2 * n / np.median(ms) / 1e6:.2f} G samples/s", flush=True)


run(3072, 4096, 16, 8, 0, "rectangle"); run(3072, 4096, 16, 8, 2, "rectangle"); run(3072, 4096, 16, 8, 2, ("tukey", 0.4))
run(768, 16384, 24, 24, 0, "rectangle"); run(768, 16384, 24, 24, 2, "rectangle"); run(768, 16384, 24, 32, 0, "rectangle")
