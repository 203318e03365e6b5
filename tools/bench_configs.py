"""Throughput of the other BASELINE.json configs (parity-test shapes, not the bench line):
device-resident subframes through flacenc_hip_qlpc_batch_async / stereo entry points."""
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from flacenc_rs_amd import _capi  # noqa: E402

import json  # noqa: E402

dev = torch.device("cuda", 0)
h = _capi.Handle(0)
ROWS = []


def run(name, frames, ch, n, bps, order, stereo):
    host = _capi.sigen_frames(frames, ch, n, bps, 200.0, 0.4, 0.1, seed=7)
    x = torch.from_numpy(host).to(dev)
    nsub = frames * (4 if stereo else ch)
    params = torch.empty((nsub, 352), dtype=torch.uint8, device=dev)
    resid = torch.empty((nsub, n), dtype=torch.int32, device=dev)
    bpsv = torch.full((nsub,), bps, dtype=torch.uint8, device=dev)
    cfg = _capi.make_config(lpc_order=order)

    def go():
        if stereo:
            h.stereo_qlpc_batch_device(cfg, x.data_ptr(), frames, n, n, bps, params.data_ptr(), resid.data_ptr(), n, stream=0)
        else:
            h.qlpc_batch_device(cfg, x.data_ptr(), frames * ch, n, n, bpsv.data_ptr(), params.data_ptr(),
                                resid.data_ptr(), n, stream=0)

    import time
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.05:  # untimed: the chip needs milliseconds of load to hold its clock
        go()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 5
    for _ in range(reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    inp = frames * ch * n
    print(f"{name:52s} {ms:8.3f} ms  {inp / ms / 1e3:9.1f} Msamples/s input  ({nsub * n / ms / 1e3:9.1f} analysed)"
          f"  {8 * inp / ms / 1e6 / 8000:6.3f} of 8 TB/s", file=sys.stderr)
    # f64 fma alone: (order + 1) per analysed sample against the FP64 vector peak (78.6 TFLOP/s = 39.3 T fma/s)
    ROWS.append({"shape": name, "frames": frames, "channels": ch, "block_size": n, "bps": bps, "lpc_order": order,
                 "ms_per_call": round(ms, 4), "Msamples_per_s_input": round(inp / ms / 1e3, 1),
                 "Msamples_per_s_analysed": round(nsub * n / ms / 1e3, 1),
                 "hbm_frac_at_8B_per_input_sample": round(8 * inp / ms / 1e6 / 8000, 4),
                 "fp64_fma_frac": round((order + 1) * nsub * n / (ms * 1e-3) / 39.3e12, 4)})


run("config1: 4096 x 16b stereo, order 10 (4 candidates)", 8192, 2, 4096, 16, 10, True)
run("config2: 4096 x 16b stereo, order 8 (4 candidates)", 8192, 2, 4096, 16, 8, True)
run("config3: 8192 x 24b stereo, order 24 (big-block kernels)", 2048, 2, 8192, 24, 24, True)
run("config3: 8192 x 24b stereo, order 32 (big-block kernels)", 2048, 2, 8192, 24, 32, True)
run("config4: 4096 x 16b 8-channel, order 10 (plain)", 2048, 8, 4096, 16, 10, False)
run("8192 x 24b stereo, order 10 (big-block kernels; the generic kernel until round 3)", 6144, 2, 8192, 24, 10, True)
run("16384 x 24b stereo, order 8 (big-block kernels; the generic kernel until round 3)", 3072, 2, 16384, 24, 8, True)
run("config5: 16384 x 24b stereo, order 24 (big-block kernels)", 1024, 2, 16384, 24, 24, True)
run("config5: 16384 x 24b stereo, order 32 (big-block kernels)", 1024, 2, 16384, 24, 32, True)
run("ragged: 4608 x 16b stereo, order 10 (72-sample-per-lane wave kernel)", 4096, 2, 4608, 16, 10, True)
run("ragged: 1152 x 16b stereo, order 8 (sub-wave kernel: 4 subframes per wave; the generic kernel until round 4)", 16384, 2, 1152, 16, 8, True)
run("ragged: 2304 x 16b stereo, order 8 (sub-wave kernel: 2 subframes per wave; the generic kernel until round 4)", 8192, 2, 2304, 16, 8, True)
run("ragged: 576 x 16b stereo, order 8 (sub-wave kernel: 8 subframes per wave)", 32768, 2, 576, 16, 8, True)
run("ragged: 1152 x 16b stereo, order 10 (sub-wave kernel)", 16384, 2, 1152, 16, 10, True)
run("2048 x 16b stereo, order 8 (sub-wave kernel)", 8192, 2, 2048, 16, 8, True)
run("1024 x 16b stereo, order 8 (sub-wave kernel)", 16384, 2, 1024, 16, 8, True)
run("256 x 16b stereo, order 8 (sub-wave kernel: 16 subframes per wave)", 65536, 2, 256, 16, 8, True)
run("1152 x 16b, 8 independent channels, order 8 (sub-wave kernel, plain batches)", 4096, 8, 1152, 16, 8, False)
# the big-block shapes at 3 x the batch: whole rounds of workgroups for every kernel (512 and 768 resident
# workgroups), launch costs amortised
run("config3, 6144 frames per launch: 8192 x 24b stereo, order 24", 6144, 2, 8192, 24, 24, True)
run("config3, 6144 frames per launch: 8192 x 24b stereo, order 32", 6144, 2, 8192, 24, 32, True)
run("config5, 3072 frames per launch: 16384 x 24b stereo, order 24", 3072, 2, 16384, 24, 24, True)
run("config5, 3072 frames per launch: 16384 x 24b stereo, order 32", 3072, 2, 16384, 24, 32, True)


def run_frames(name, frames, ch, n, bps, order):
    """Frame-level pipeline for Independent(ch) frames: encode_frames (+ fixed-LPC) and pack_frames."""
    host = _capi.sigen_frames(frames, ch, n, bps, 200.0, 0.4, 0.1, seed=7)
    x = torch.from_numpy(host).to(dev)
    res = torch.empty((frames * ch, 368), dtype=torch.uint8, device=dev)
    resid = torch.empty((frames * ch, n), dtype=torch.int32, device=dev)
    stride = int(h._lib.flacenc_hip_frame_bytes_bound(ch, n, bps))
    out = torch.empty((frames, stride), dtype=torch.uint8, device=dev)
    lens = torch.zeros(frames, dtype=torch.int32, device=dev)
    cfg = _capi.make_frame_config(_capi.make_config(lpc_order=order), use_fixed=True)

    def go():
        h._check(h._lib.flacenc_hip_encode_frames_async(h._h, cfg, x.data_ptr(), frames, ch, n, n, bps, res.data_ptr(),
                                                        resid.data_ptr(), n, None))
        h._check(h._lib.flacenc_hip_pack_frames_async(h._h, x.data_ptr(), frames, ch, n, n, res.data_ptr(),
                                                      resid.data_ptr(), n, bps, 44100, 0, 1, out.data_ptr(), stride,
                                                      lens.data_ptr(), None))

    for _ in range(2):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        go()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    inp = frames * ch * n
    print(f"{name:52s} {ms:8.3f} ms  {inp / ms / 1e3:9.1f} Msamples/s input, PCM -> frame bytes "
          f"({int(lens.sum().item()) / (inp * bps / 8):.3f} of PCM size)", file=sys.stderr)
    ROWS.append({"shape": name, "frames": frames, "channels": ch, "block_size": n, "bps": bps, "lpc_order": order,
                 "ms_per_call": round(ms, 4), "Msamples_per_s_input": round(inp / ms / 1e3, 1),
                 "what": "encode_frames (default config: fixed-LPC candidate on) + pack_frames, PCM in HBM -> frame bytes in HBM"})


def run_stereo_frames(name, frames, n, bps, order):
    """flacenc_hip_encode_stereo_frames with the reference's default candidate set (fixed-LPC on), device-resident."""
    host = _capi.sigen_frames(frames, 2, n, bps, 200.0, 0.4, 0.1, seed=7)
    x = torch.from_numpy(host).to(dev)
    res = torch.empty((frames, 752), dtype=torch.uint8, device=dev)
    resid = torch.empty((frames * 2, n), dtype=torch.int32, device=dev)
    cfg = _capi.make_frame_config(_capi.make_config(lpc_order=order), use_fixed=True)

    def go():
        h.encode_stereo_frames_device(cfg, x.data_ptr(), frames, n, n, bps, res.data_ptr(), resid.data_ptr(), n, stream=0)

    import time
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.05:
        go()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        go()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    inp = frames * 2 * n
    print(f"{name:52s} {ms:8.3f} ms  {inp / ms / 1e3:9.1f} Msamples/s input, frame decisions", file=sys.stderr)
    ROWS.append({"shape": name, "frames": frames, "channels": 2, "block_size": n, "bps": bps, "lpc_order": order,
                 "ms_per_call": round(ms, 4), "Msamples_per_s_input": round(inp / ms / 1e3, 1),
                 "what": "encode_stereo_frames, default candidates (QLPC + fixed-LPC for L, R, M, S, encode_frame's decision, two rows out)"})


run_stereo_frames("frames: 1152 x 16b stereo, default candidates, order 8 (sub-wave kernel, one launch)", 16384, 1152, 16, 8)
run_stereo_frames("frames: 2304 x 16b stereo, default candidates, order 8 (sub-wave kernel, one launch)", 8192, 2304, 16, 8)
run_stereo_frames("frames: 1152 x 16b stereo, default candidates, order 10 (sub-wave kernel, one launch)", 16384, 1152, 16, 10)
run_stereo_frames("frames: 2048 x 16b stereo, default candidates, order 8 (sub-wave kernel, one launch)", 8192, 2048, 16, 8)
run_stereo_frames("frames: 256 x 16b stereo, default candidates, order 8 (sub-wave kernel, one launch)", 65536, 256, 16, 8)
run_stereo_frames("frames: 4096 x 16b stereo, default candidates, order 10 (fused wave kernel)", 8192, 4096, 16, 10)
run_frames("config4: 4096 x 16b 8-channel, default config, frames", 2048, 8, 4096, 16, 10)
run_frames("mono: 4096 x 16b, default config, frames", 8192, 1, 4096, 16, 10)
run_frames("1152 x 16b 8-channel, default config, frames (sub-wave kernel, one launch + the packer)", 2048, 8, 1152, 16, 10)

print(json.dumps({"tool": "tools/bench_configs.py", "timing": "HIP events around 5 calls after 2 warm-up calls and 50 ms of untimed clock spin-up, device-resident data",
                  "rows": ROWS}, indent=1))
