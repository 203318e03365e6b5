"""Blocks of 576 / 1152 / 2304 samples on the generic kernel: ms per call and input samples/s."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from flacenc_rs_amd import _capi  # noqa: E402

dev = torch.device("cuda", 0)
h = _capi.Handle(0)
for n, frames in ((576, 32768), (1152, 16384), (2304, 8192)):
    for order in (8, 10):
        host = _capi.sigen_frames(frames, 2, n, 16, 200.0, 0.4, 0.1, seed=7)
        x = torch.from_numpy(host).to(dev)
        params = torch.empty((frames * 4, 352), dtype=torch.uint8, device=dev)
        resid = torch.empty((frames * 4, n), dtype=torch.int32, device=dev)
        cfg = _capi.make_config(lpc_order=order)

        def go():
            h.stereo_qlpc_batch_device(cfg, x.data_ptr(), frames, n, n, 16, params.data_ptr(), resid.data_ptr(), n, stream=0)

        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.05:
            go()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            go()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"n={n} order={order} threads={os.environ.get('FLACENC_EXP_THREADS', 'plan')}: {ms:.3f} ms  {frames * 2 * n / ms / 1e6:.1f} G samples/s")
