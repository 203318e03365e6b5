"""Time the reference-order autocorrelation pre-pass alone (R[] only): qlpc batch in reference-order mode minus ...
Simplest honest figure: rocprofv3-free event timing of encode_stereo_frames with the flag, printed next to the
canonical order, for the library named by FLACENC_HIP_LIB."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from flacenc_rs_amd import _capi
F, n, bps = (int(sys.argv[2]) if len(sys.argv) > 2 else 24576), 4096, 16
order = int(sys.argv[1]) if len(sys.argv) > 1 else 10
h = _capi.Handle(0)
x = torch.from_numpy(_capi.sigen_frames(F, 2, n, bps, 200.0, 0.4, 0.4, seed=1)).cuda()
res = torch.empty((F, 752), dtype=torch.uint8, device="cuda")
rr = torch.empty((F * 2, n), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream()
out = []
for name, flags, fixed in (("canonical", 0, False), ("reference", _capi.FLAG_REFERENCE_SUM_ORDER, False),
                           ("reference+fixed", _capi.FLAG_REFERENCE_SUM_ORDER, True)):
    cfg = _capi.make_frame_config(_capi.make_config(lpc_order=order, flags=flags), use_fixed=fixed)
    ts = []
    for i in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        h.encode_stereo_frames_device(cfg, x.data_ptr(), F, n, n, bps, res.data_ptr(), rr.data_ptr(), n, stream=st.cuda_stream)
        e1.record(st)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[4:])
    out.append(f"{name} {ts[len(ts)//2]:.3f} ms")
print(os.path.basename(os.environ.get("FLACENC_HIP_LIB", "default")), f"order {order}:", "; ".join(out))
