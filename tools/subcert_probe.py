import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from flacenc_rs_amd import _capi
dev = torch.device("cuda", 0)
h = _capi.Handle(0, hooks=True)
for n, frames in ((1152, 16384),):
    for noise in ((0.1,) if os.environ.get("PROBE_ONE") else (0.4, 0.1)):
        for order in (8,):
            host = _capi.sigen_frames(frames, 2, n, 16, 200.0, 0.4, noise, seed=7)
            x = torch.from_numpy(host).to(dev)
            params = torch.empty((frames * 4, 352), dtype=torch.uint8, device=dev)
            resid = torch.empty((frames * 4, n), dtype=torch.int32, device=dev)
            for flags in ((0,) if os.environ.get("PROBE_ONE") else (0, _capi.FLAG_CANONICAL_SUM_ORDER)):
                cfg = _capi.make_config(lpc_order=order, flags=flags)
                go = lambda: h.stereo_qlpc_batch_device(cfg, x.data_ptr(), frames, n, n, 16, params.data_ptr(), resid.data_ptr(), n, stream=0)
                for _ in range(20): go()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): go()
                e1.record(); torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 10
                st = torch.zeros(3, dtype=torch.int32, device=dev)
                h.debug_set_cert_stats(st.data_ptr()); go(); torch.cuda.synchronize(); h.debug_set_cert_stats(0)
                print(f"n={n} noise={noise} order={order} flags={flags}: {ms:.3f} ms {frames*2*n/ms/1e6:.1f} G  stats {st.cpu().tolist()}")
