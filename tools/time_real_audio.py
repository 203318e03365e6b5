"""The headline kernel on stereo frames cut from the reference's real-audio fixtures (tests/golden), tiled to a batch:
rate and certificate counters per order, with and without the certificate (FLACENC_HIP_FLAG_CANONICAL_SUM_ORDER).
    gpurun -- python tools/time_real_audio.py [frames=24576]"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from flacenc_rs_amd import _capi

F = int(sys.argv[1]) if len(sys.argv) > 1 else 24576
n, bps = 4096, 16
gold = os.path.join(ROOT, "tests", "golden")
cut = []
for nm in ("sus109", "sus6", "ras22", "ras103"):
    ch = [np.fromfile(os.path.join(gold, "testsignal.%s.ch%d.bin" % (nm, c)), dtype="<i2").astype(np.int32) for c in (0, 1)]
    for t0 in range(0, 8192 - n + 1, 64):
        cut.append(np.stack([ch[0][t0:t0 + n], ch[1][t0:t0 + n]]))
cut = np.stack(cut)
x = torch.from_numpy(np.ascontiguousarray(np.tile(cut, ((F + len(cut) - 1) // len(cut), 1, 1))[:F])).cuda()
results = torch.zeros((F, _capi.FRAME_RESULT_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
residual = torch.zeros((F * 2, n), dtype=torch.int32, device="cuda")
h = _capi.Handle(0, hooks=True)
for order in (8, 10, 12):
    for adaptive, flags in ((1, 0), (0, 0), (0, _capi.FLAG_CANONICAL_SUM_ORDER)):
        h.debug_set_adaptive_order(bool(adaptive))
        cfg = _capi.make_frame_config(_capi.make_config(lpc_order=order, flags=flags), use_fixed=False)
        go = lambda: h.encode_stereo_frames_device(cfg, x.data_ptr(), F, n, n, bps, results.data_ptr(), residual.data_ptr(), n, stream=0)
        for _ in range(5):
            go()
            torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(18)]
        for a, b in ev:
            a.record(); go(); b.record()
        torch.cuda.synchronize()
        t = [a.elapsed_time(b) for a, b in ev]
        state = h.debug_adaptive_state()
        print("   per launch:", " ".join("%.2f" % v for v in t))
        st = torch.zeros(3, dtype=torch.int32, device="cuda")
        h.debug_set_cert_stats(st.data_ptr()); go(); torch.cuda.synchronize(); h.debug_set_cert_stats(0)
        print("order %2d adaptive %d flags %3d: mean %.3f ms (median %.3f)  %.1f G samples/s  cert stats %s  state %s" % (
            order, adaptive, flags, float(np.mean(t)), float(np.median(t)), F * 2 * n / float(np.mean(t)) / 1e6, st.cpu().tolist(), state))
