"""Where do the waves of the fused kernel's workgroups run?  Diagnostic build -DFLACENC_STAMP_HWID (stamp slot 7 =
HW_REG_HW_ID | HW_REG_XCC_ID << 32): SIMD of each wave index, and -- for workgroups resident on one CU at the same time --
how many of their waves 0 share a SIMD.   FLACENC_HIP_LIB=ab/libflacenc_hip_hwid.so python tools/hwid_probe.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from flacenc_rs_amd import _capi
F, n, bps = 8192, 4096, 16
x = torch.from_numpy(_capi.sigen_frames(F, 2, n, bps, 200.0, 0.4, 0.4, seed=1)).cuda()
results = torch.zeros((F, 752), dtype=torch.uint8, device="cuda"); residual = torch.zeros((F * 2, n), dtype=torch.int32, device="cuda")
st = torch.zeros((F * 4, 8), dtype=torch.int64, device="cuda")
h = _capi.Handle(0, hooks=True); h.debug_set_stamps(st.data_ptr())
cfg = _capi.make_frame_config(_capi.make_config(lpc_order=8), use_fixed=False)
for _ in range(2):
    h.encode_stereo_frames_device(cfg, x.data_ptr(), F, n, n, bps, results.data_ptr(), residual.data_ptr(), n, stream=0)
torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.uint64).reshape(F, 4, 8)
hw = (s[:, :, 7] & np.uint64(0xFFFFFFFF)).astype(np.uint32); xcc = (s[:, :, 7] >> np.uint64(32)).astype(np.uint32) & 0xF
wave_id, simd, cu, sh, se = hw & 0xF, (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 12) & 1, (hw >> 13) & 7
print("SIMD histogram by wave index (rows: wave 0..3, columns: SIMD 0..3)")
for w in range(4):
    print("  wave", w, np.bincount(simd[:, w], minlength=4).tolist())
print("distinct SIMDs among a workgroup's four waves:", np.bincount([len(set(simd[f])) for f in range(F)], minlength=5).tolist())
# co-residency: workgroups on the same CU, overlapping in time (stamp 0 .. stamp 6 of wave 0)
key = (xcc[:, 0].astype(np.int64) << 12) | (se[:, 0].astype(np.int64) << 8) | (sh[:, 0].astype(np.int64) << 4) | cu[:, 0]
t0, t1 = s[:, 0, 0].astype(np.int64), s[:, 0, 6].astype(np.int64)
same = tot = 0
for k in np.unique(key)[:64]:
    idx = np.nonzero(key == k)[0]
    idx = idx[np.argsort(t0[idx])]
    for a in range(len(idx)):
        for b in range(a + 1, len(idx)):
            if t0[idx[b]] < t1[idx[a]]:
                tot += 1; same += int(simd[idx[a], 0] == simd[idx[b], 0])
print("pairs of co-resident workgroups on a CU: %d, of which waves 0 on the same SIMD: %d (%.2f)" % (tot, same, same / max(tot, 1)))
print("CUs seen:", len(np.unique(key)), " example hw_id of frame 0:", [hex(int(v)) for v in hw[0]], "xcc", xcc[0].tolist())
