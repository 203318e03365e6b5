"""Debug aid: one seed of tools/fuzz_subwave.py, the independent-channel frame call against the candidate batch and the oracle.
    gpurun -- python tools/debug_subwave_seed.py <seed> <subframe>"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from flacenc_rs_amd import _capi
from oracle import oracle as orc
src = open(os.path.join(ROOT, "tools", "fuzz_subwave.py")).read()
ns = {"__file__": os.path.join(ROOT, "tools", "fuzz_subwave.py")}
exec(compile(src[:src.index("def _diff_channels")], "fz", "exec"), ns)
signals, SIZES = ns["signals"], ns["SIZES"]
seed, sfi = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(770000 + seed)
n = int(rng.choice(SIZES)); bps = int(rng.choice([8, 12, 16, 16, 16, 20, 24, 24])); order = int(rng.integers(1, 13))
qkw = dict(lpc_order=order, quant_precision=int(rng.integers(2, 16)),
           window=("rectangle" if rng.random() < 0.2 else ("tukey", float(np.round(rng.random(), 2)))),
           max_rice_parameter=int(rng.choice([0, 3, 7, 14, 15, 30, 30])), rice_finest_only=bool(rng.random() < 0.1))
fkw = dict(use_constant=bool(rng.random() < 0.85), use_fixed=bool(rng.random() < 0.75), use_lpc=bool(rng.random() < 0.9),
           use_leftside=bool(rng.random() < 0.8), use_rightside=bool(rng.random() < 0.8), use_midside=bool(rng.random() < 0.8),
           fixed_max_order=int(rng.integers(0, 5)), fixed_partitions=int(rng.choice([1, 2, 4, 8, 16, 16, 32, 64, 3, 12])))
nf = int(rng.integers(1, 20))
x = signals(rng, nf, n, bps)
flat = x.reshape(nf * 2, n)
bpsv = rng.choice([bps, bps, min(bps + 1, 25)], nf * 2).astype(np.uint8)
chn = int(rng.choice([1, 3, 8])); nfc = (nf * 2) // chn
xc = flat[: nfc * chn].reshape(nfc, chn, n)
print(seed, n, bps, qkw, fkw, "channels", chn, "frames", nfc)
h = _capi.Handle(0, hooks=True)
q0 = _capi.make_config(**qkw)
st = torch.zeros(3, dtype=torch.int32, device="cuda")
for name, call in (("plain candidates", lambda: h.qlpc_batch(flat[: nfc * chn], bps, q0)),
                   ("encode_frames", lambda: h.encode_frames(xc, bps, _capi.make_frame_config(q0, **fkw))),
                   ("encode_frames bps+1", lambda: h.encode_frames(xc, bps + 1, _capi.make_frame_config(q0, **fkw))),
                   ("encode_frames/2", lambda: h.encode_frames(xc // 2, bps, _capi.make_frame_config(q0, **fkw))),
                   ("plain candidates/2", lambda: h.qlpc_batch(flat[: nfc * chn] // 2, bps, q0))):
    st.zero_(); h.debug_set_cert_stats(st.data_ptr()); out = call(); torch.cuda.synchronize(); h.debug_set_cert_stats(0)
    rec = out[0].reshape(-1)[sfi]
    p = rec["params"] if "params" in (rec.dtype.names or ()) else rec
    print("%-18s cert stats %s  subframe %d: coefs %s shift %d status %d bits %d" % (
        name, st.cpu().tolist(), sfi, p["coefs"][:order], p["shift"], p["status"], p["subframe_bits"]))
kw = dict(lpc_order=order, quant_precision=qkw["quant_precision"], window=qkw["window"])
orc.cert_stats(reset=True)
cp, _, _, _ = orc.qlpc_batch(flat[: nfc * chn], bps, orc.make_config(acorr=orc.ACORR_CANONICAL, **kw), nthreads=1)
print("oracle             cert stats %s  subframe %d: coefs %s" % (list(orc.cert_stats()), sfi, cp["coefs"][sfi][:order]))
