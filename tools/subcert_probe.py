"""The sub-wave kernel's shapes unflagged (two passes: the reference's chains in front) and with
FLACENC_HIP_FLAG_CANONICAL_SUM_ORDER (one pass, the chunk tree): the stereo candidate batch (four subframes per frame) per
block size, order and material -- the bench signal and stereo frames cut from the reference's real-audio fixtures.
    gpurun -- python tools/subcert_probe.py [samples per launch = 2^25] [block sizes = 256,576,1152,2304]
Prints per row: ms and G samples/s of both launches.  (Round 6 first built an order certificate inside the kernel and used
this probe to price it -- its counters were the last column; the unflagged order on these shapes is now the reference's by
two passes and the counters stay at zero: profiles/r06_subwave_two_pass.txt.)"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from flacenc_rs_amd import _capi

TOTAL = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 25
dev = torch.device("cuda", 0)
h = _capi.Handle(0, hooks=True)
gold = os.path.join(ROOT, "tests", "golden")
fixtures = [[np.fromfile(os.path.join(gold, "testsignal.%s.ch%d.bin" % (nm, c)), dtype="<i2").astype(np.int32) for c in (0, 1)]
            for nm in ("sus109", "sus6", "ras22", "ras103")]


def music(frames, n):
    cut = []
    for ch in fixtures:
        for t0 in range(0, 8192 - n + 1, 64):
            cut.append(np.stack([ch[0][t0:t0 + n], ch[1][t0:t0 + n]]))
    cut = np.stack(cut)
    return np.ascontiguousarray(np.tile(cut, ((frames + len(cut) - 1) // len(cut), 1, 1))[:frames])


def timed(go, reps=10):
    for _ in range(10):
        go()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); go(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


for n in ([int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else (256, 576, 1152, 2304)):
    frames = TOTAL // (2 * n)
    for material in ("bench signal", "real audio"):
        host = _capi.sigen_frames(frames, 2, n, 16, 200.0, 0.4, 0.4, seed=7) if material == "bench signal" else music(frames, n)
        x = torch.from_numpy(host).to(dev)
        params = torch.empty((frames * 4, 352), dtype=torch.uint8, device=dev)
        resid = torch.empty((frames * 4, n), dtype=torch.int32, device=dev)
        for order in (8, 12):
            row = []
            for flags in (0, _capi.FLAG_CANONICAL_SUM_ORDER):
                cfg = _capi.make_config(lpc_order=order, flags=flags)
                go = lambda: h.stereo_qlpc_batch_device(cfg, x.data_ptr(), frames, n, n, 16, params.data_ptr(), resid.data_ptr(), n, stream=0)
                ms = timed(go)
                row.append((ms, frames * 2 * n / ms / 1e6))
                if flags == 0:
                    st = torch.zeros(3, dtype=torch.int32, device=dev)
                    h.debug_set_cert_stats(st.data_ptr()); go(); torch.cuda.synchronize(); h.debug_set_cert_stats(0)
                    stats = st.cpu().tolist()
            print("n %4d order %2d %-12s unflagged (two passes) %.3f ms %6.1f G | CANONICAL_SUM_ORDER (one pass) %.3f ms %6.1f G | x %.3f | analysed / second tier / marked %s (%.2f %% / %.2f %%)" % (
                n, order, material, row[0][0], row[0][1], row[1][0], row[1][1], row[0][0] / row[1][0], stats,
                100.0 * stats[1] / max(1, stats[0]), 100.0 * stats[2] / max(1, stats[0])), flush=True)
        del x, params, resid

