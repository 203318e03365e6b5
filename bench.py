#!/usr/bin/env python3
"""bench.py -- throughput of the QLPC analysis hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic stereo frames that
are already resident in HBM: per frame the four `estimated_qlpc` analyses encode_frame
makes (L, R, M, S; src/coding.rs:530-544, 476-491), each = window -> f64 autocorrelation
-> Levinson -> quantisation -> integer residual -> exhaustive partitioned-Rice search.
Workload = BASELINE.json configs[1]: 44.1 kHz / 16-bit stereo, block 4096, LPC order 8
(the reference has no "fixed Rice partition order" mode, so the reference-faithful full
search is what runs).  value = input channel-samples (frames x 2 x 4096) per second over
all ranks.  Multi-GPU: frames are sharded over ranks (weak scaling); each step every rank derives
its frames' byte lengths from the decision records and the ranks all-gather those 4 bytes per
frame over RCCL and prefix-sum them into stream offsets (ParSink's ordered gather, src/par.rs:67-95,
reduced to what ordering needs); records and residuals stay on the producing GPU.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALGO_BYTES_PER_SAMPLE = 8.0  # 4 B sample read + 4 B residual written per input channel-sample


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=8192, help="stereo frames per step per GPU")
    ap.add_argument("--block-size", type=int, default=4096)
    ap.add_argument("--lpc-order", type=int, default=8)
    ap.add_argument("--bps", type=int, default=16)
    ap.add_argument("--use-fixed", action="store_true",
                    help="also run the fixed-LPC candidate (the reference's default SubFrameCoding); "
                         "not the north-star workload, reported in DESIGN.md")
    ap.add_argument("--finest-rice-order", action="store_true",
                    help="build extension (not a reference mode): keep the finest Rice partition order, "
                         "BASELINE config 2's 'fixed Rice partition order'; reported in DESIGN.md, not the default")
    ap.add_argument("--gather", choices=["lengths", "records"], default="lengths",
                    help="what the multi-GPU exchange step moves: the frames' byte lengths (4 B/frame, enough to "
                         "place every frame in the stream; default) or the whole 752-B decision records "
                         "(every SubFrame component on every rank)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the multi-GPU exchange step (frame lengths -> offsets) even with one rank; "
                         "for measuring its cost, not a reported configuration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU work for cpu_baseline")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from flacenc_rs_amd import _capi, shard

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    n, F, bps = args.block_size, args.frames, args.bps
    # precision 15, Tukey(0.4), max_p 30; candidates Constant / Verbatim / LPC -- the QLPC analysis path
    # the metric names (--use-fixed adds the reference default's fixed-LPC candidate); all stereo
    # assignments allowed
    cfg = _capi.make_frame_config(_capi.make_config(lpc_order=args.lpc_order, rice_finest_only=args.finest_rice_order),
                                  use_fixed=args.use_fixed)
    # synthetic "sigen" audio: Sine(200, 0.4) + Noise(0.4) like the reference's
    # stereo_frame_encoder_noisy_sine_lpc bench (src/coding.rs:1152), one continuous stream,
    # dealt round-robin: stream frame f belongs to rank f mod G (flacenc_rs_amd/shard.py)
    host = _capi.sigen_frames(F, 2, n, bps, 200.0, 0.4, 0.4, seed=0xF1AC0001, first_frame=rank,
                              frame_step=world)
    x = torch.from_numpy(host).to(dev)
    rec_bytes = _capi.FRAME_RESULT_DTYPE.itemsize
    # decision records and frame lengths are double-buffered: the exchange of step k (on its own
    # stream) overlaps the analysis of step k + 1
    results2 = [torch.empty((F, rec_bytes), dtype=torch.uint8, device=dev) for _ in range(2)]
    frame_len2 = [torch.zeros(F, dtype=torch.int32, device=dev) for _ in range(2)]
    results = results2[0]
    residual = torch.empty((F * 2, n), dtype=torch.int32, device=dev)     # the two chosen channels
    handle = _capi.Handle(local_rank)
    stream = torch.cuda.current_stream()
    comm = torch.cuda.Stream(device=dev)
    exchanging = world > 1 or args.force_exchange
    consumed = [None, None]  # event: the exchange that read buffer b has finished
    step_no = [0]

    def exchange(b):
        # the multi-GPU exchange step: frame byte lengths -> stream order -> stream offsets
        handle.stereo_frame_lengths_device(results2[b].data_ptr(), F, n, bps, 44100, rank, world,
                                           frame_len2[b].data_ptr(), stream=comm.cuda_stream)
        lengths_all = shard.all_gather_frame_lengths(frame_len2[b], world * F)
        if args.gather == "records":
            shard.all_gather_records(results2[b], world * F)
        return shard.stream_offsets(lengths_all)[0]

    def step(events=None):
        b = step_no[0] & 1
        step_no[0] += 1
        if exchanging and consumed[b] is not None:
            stream.wait_event(consumed[b])  # the records of two steps ago have been read
        if events:
            events[0].record(stream)
        handle.encode_stereo_frames_device(cfg, x.data_ptr(), F, n, n, bps, results2[b].data_ptr(),
                                           residual.data_ptr(), n, stream=stream.cuda_stream)
        if events:
            events[1].record(stream)
        if exchanging:
            ready = torch.cuda.Event()
            ready.record(stream)
            with torch.cuda.stream(comm):
                comm.wait_event(ready)
                exchange(b)
                consumed[b] = torch.cuda.Event()
                consumed[b].record(comm)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # per-launch kernel time: HIP events recorded on the stream the kernel is launched on
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(ev[k])
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) if args.steps else float("nan")

    # sanity: nothing in the timed region may have failed
    p = np.frombuffer(results2[(step_no[0] - 1) & 1].cpu().numpy().tobytes(), dtype=_capi.FRAME_RESULT_DTYPE)
    lpc_kind = p["kind"] >= 2
    assert (p["lpc"]["status"][lpc_kind] == 0).all(), "subframe status != 0"
    chosen_bits = int(sum(int(p["bits"][f, r]) for f in range(F) for r in p["role"][f]))
    assign_hist = np.bincount(p["channel_assignment"], minlength=4).tolist()

    samples_per_step = F * 2 * n  # input channel-samples per rank per step
    value = world * samples_per_step * args.steps / elapsed / 1e6
    achieved = ALGO_BYTES_PER_SAMPLE * samples_per_step / (kernel_ms * 1e-3) / 1e9

    out = {
        "metric": "Msamples/s encoded (44.1kHz/16b stereo, block=4096): QLPC analysis path",
        "value": round(value, 2),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64+int32",
        "data": "synthetic",
        "config": {
            "workload": "configs[1]: sigen Sine(200,0.4)+Noise(0.4), 44.1kHz/16-bit stereo, "
                        f"block_size={n}, LPC order {args.lpc_order}, precision 15, Tukey(0.4), "
                        + ("finest Rice partition order only (build extension)" if args.finest_rice_order
                           else "full partitioned-Rice search") +
                        " (max_p 30); L,R,M,S analysed per frame, encode_frame decision",
            "frames_per_step_per_gpu": F,
            "subframes_analysed_per_step_per_gpu": 4 * F,
            "decision": ("encode_subframe {Constant, Verbatim, FixedLpc(ApproxEnt 16), LPC}" if args.use_fixed
                         else "encode_subframe {Constant, Verbatim, LPC}") +
                        " + try_stereo_coding on the GPU; the two chosen residuals written",
            "gather": (("all_gather of per-frame byte lengths (4 B/frame, RCCL) + prefix sum to stream offsets"
                        + (" + all_gather of the 752-B frame records" if args.gather == "records" else "")
                        + ", on its own stream, overlapping the next step's analysis") if world > 1 else "none"),
            "subframe_bits_per_sample": round(chosen_bits / (2 * F * n), 4),
            "assignments_indep_left_right_mid": assign_hist,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": _measured_traffic(),
            "kernel": "qlpc_wave4096_kernel<8,true,true,%s,false>" % ("true" if args.use_fixed else "false"),
            "kernel_ms": round(kernel_ms, 4),
            "algorithmic_bytes_per_launch": ALGO_BYTES_PER_SAMPLE * samples_per_step,
        },
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(host, bps, args, n)
    if rank == 0:
        print(json.dumps(out), flush=True)
    handle.close()
    if world > 1:
        dist.destroy_process_group()


def _measured_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/), or None."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get("bytes_per_launch")
    except Exception:
        return None


def usable_cores():
    """Host threads this process may really use: affinity mask capped by the cgroup CPU quota."""
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except Exception:
        pass
    return cores


def cpu_baseline(host, bps, args, n):
    """The oracle (a port of the reference's path in reference summation order) timed on this
    box's host cores over a bounded sample of the same frames."""
    from oracle import oracle as orc

    cores = usable_cores()
    ocfg = orc.make_config(lpc_order=args.lpc_order)
    sample_frames = min(host.shape[0], 16 * cores)
    sample = np.ascontiguousarray(host[:sample_frames])
    secs, _ = orc.bench_stereo_qlpc(sample, bps, ocfg, cores, 1)  # calibration pass
    repeats = max(1, int(args.cpu_seconds / max(secs, 1e-3)))
    secs, _ = orc.bench_stereo_qlpc(sample, bps, ocfg, cores, repeats)
    v = sample_frames * 2 * n * repeats / secs / 1e6
    return {
        "value": round(v, 2),
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{sample_frames} of the same stereo frames x {repeats} passes, {cores} threads "
                  "(frame-parallel pool like src/par.rs), 4 analyses per frame",
    }


if __name__ == "__main__":
    main()
