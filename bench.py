#!/usr/bin/env python3
"""bench.py -- throughput of the QLPC analysis hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic stereo frames that
are already resident in HBM: per frame the four `estimated_qlpc` analyses encode_frame
makes (L, R, M, S; src/coding.rs:530-544, 476-491), each = window -> f64 autocorrelation
-> Levinson -> quantisation -> integer residual -> exhaustive partitioned-Rice search.
Workload = BASELINE.json configs[1]: 44.1 kHz / 16-bit stereo, block 4096, LPC order 8
(the reference has no "fixed Rice partition order" mode, so the reference-faithful full
search is what runs).  value = input channel-samples (frames x 2 x 4096) per second over
all ranks.

Multi-GPU: one process per GPU, stream frame f belongs to rank f mod G (weak scaling: fixed frames
per GPU).  Each step every rank derives its frames' byte lengths from the decision records, and the
ranks all-gather over RCCL (a) those lengths and (b) the encoded SubFrame components -- the 752-byte
frame records: channel assignment, subframe kinds, quantised coefficients, shift, Rice partition
order and parameters, bit counts -- and put both into stream order (ParSink's ordered gather,
src/par.rs:67-95).  `--gather payload` moves the packed FLAC frame bytes as well, `--gather lengths`
only what ordering needs.  The exchange of step k runs on its own HIP stream under step k + 1.

    python bench.py                         # 1 GPU
    python bench.py --gpus 8                # starts 8 ranks itself (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N   # or let a launcher start the ranks
    python bench.py --gpus 2 --backend gloo --dry-run   # CPU check of launch + exchange plumbing
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALGO_BYTES_PER_SAMPLE = 8.0  # 4 B sample read + 4 B residual written per input channel-sample
N_SIMDS = 256 * 4  # 256 CUs x 4 SIMDs
SAMPLE_RATE = 44100


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=98304,
                    help="stereo frames per step per GPU: 128 full rounds of the 768 workgroups an MI355X holds, "
                         "3.2 GB of samples in and 3.2 GB of residual out per step (sized for 288 GB of HBM; a launch "
                         "has ~16 us of fixed cost and a partial last round, tools/launch_cost.py: 8192 frames "
                         "measure 295, 24576 frames 315, 98304 frames 326 G samples/s on the same kernel)")
    ap.add_argument("--block-size", type=int, default=4096)
    ap.add_argument("--lpc-order", type=int, default=8)
    ap.add_argument("--bps", type=int, default=16)
    ap.add_argument("--use-fixed", action="store_true",
                    help="also run the fixed-LPC candidate (the reference's default SubFrameCoding); "
                         "not the north-star workload (it is reported under `secondary`)")
    ap.add_argument("--finest-rice-order", action="store_true",
                    help="build extension (not a reference mode): keep the finest Rice partition order, "
                         "BASELINE config 2's 'fixed Rice partition order'; not the default")
    ap.add_argument("--reference-order", action="store_true",
                    help="FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER: autocorrelation in the reference's stable "
                         "summation order (bit-identical coefficients; one extra pass over the samples)")
    ap.add_argument("--canonical-order", action="store_true",
                    help="FLACENC_HIP_FLAG_CANONICAL_SUM_ORDER: the bare chunk tree, without the order certificate that makes "
                         "the default mode's integers the reference's (A/B of the certificate's cost; not the default)")
    ap.add_argument("--gather", choices=["records", "payload", "lengths"], default="records",
                    help="what the multi-GPU exchange moves besides the frames' byte lengths: the 752-B "
                         "decision records = the encoded SubFrame components (default), the packed frame "
                         "bytes too (payload), or nothing (lengths)")
    ap.add_argument("--no-stream-priority", action="store_true",
                    help="exchange runs: leave the analysis on a normal-priority stream (A/B of the priority's effect)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the multi-GPU exchange step even with one rank; for measuring its cost")
    ap.add_argument("--comm", choices=["torch", "capi"], default="torch",
                    help="who runs the exchange's all-gathers: torch.distributed (default) or the C library's own RCCL "
                         "communicator (flacenc_hip_comm_create + flacenc_hip_allgather_records_async: what a Rust / C++ "
                         "host with one process per GPU calls); torch.distributed stays for the launch, the barrier and "
                         "the timing reduction, and hands rank 0's communicator id to the other ranks")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend; nccl is RCCL on ROCm.  gloo only with --dry-run")
    ap.add_argument("--shared-gpu-test", action="store_true",
                    help="functional test of the real multi-rank path on a box with fewer GPUs than ranks: ranks "
                         "share devices (rank mod device count) and the collectives run over gloo.  Not a "
                         "measurement: the line is marked and `value` is 0")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU check of the launch and exchange plumbing (no GPU, no analysis, no "
                         "measurement): stand-in records, real sharding + collectives over gloo")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU work for cpu_baseline")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    """`--gpus N` without a launcher: start N rank processes (one per GPU) and relay rank 0's JSON
    line.  This parent never imports torch or touches the GPU, and nothing is exec'ed from a process
    that has: the ranks are fresh children of torch.distributed.run."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.backend == "gloo" and not (args.dry_run or args.shared_gpu_test):
        raise SystemExit("--backend gloo is only for --dry-run / --shared-gpu-test (the product path has no CPU fallback)")
    if args.shared_gpu_test and args.backend != "gloo":
        raise SystemExit("--shared-gpu-test needs --backend gloo (RCCL wants one device per rank)")
    if args.comm == "capi" and (args.dry_run or args.shared_gpu_test or args.backend == "gloo"):
        raise SystemExit("--comm capi is the library's RCCL communicator: one GPU per rank, no gloo / dry-run form")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args, argv))
    world = int(env_world or "1")
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; start one rank per GPU "
              f"(python bench.py --gpus N does that itself)", file=sys.stderr)
        sys.exit(2)
    if args.dry_run:
        dry_run(args, world)
    else:
        run(args, world)


# ---------------------------------------------------------------------------------------------
def kernel_source_sha():
    """Fingerprint of the fused kernel's source: profile-derived figures are only attached to a run of
    the very code they were measured on."""
    h = hashlib.sha256()
    for name in ("qlpc_wave_kernel_impl.h", "qlpc_kernel_impl.h", "qlpc_wave_inst.hip", "qlpc_dispatch.cpp"):
        with open(os.path.join(ROOT, "flacenc_rs_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def profiled_counters(args):
    """Per-launch figures from the committed rocprofv3 PMC passes (profiles/headline_pmc.json): HBM
    bytes and VALU instructions per wave.  None unless this run is the profiled configuration AND the
    kernel source is the profiled one."""
    default = parse_args([])
    same_cfg = all(getattr(args, k) == getattr(default, k)
                   for k in ("frames", "block_size", "lpc_order", "bps", "use_fixed", "finest_rice_order",
                             "reference_order"))
    try:
        with open(os.path.join(ROOT, "profiles", "headline_pmc.json")) as f:
            p = json.load(f)
    except Exception:
        return None
    if not same_cfg or p.get("kernel_source_sha") != kernel_source_sha():
        return None
    return p


def profiled_phases(args):
    """Per-phase VALU instructions of the headline kernel next to the formulation's floor (profiles/
    headline_phases.json, tools/phase_insts.sh): attached like the counters, to the profiled source only."""
    default = parse_args([])
    same_cfg = all(getattr(args, k) == getattr(default, k)
                   for k in ("block_size", "lpc_order", "bps", "use_fixed", "finest_rice_order", "reference_order"))
    try:
        with open(os.path.join(ROOT, "profiles", "headline_phases.json")) as f:
            p = json.load(f)
    except Exception:
        return None
    if not same_cfg or p.get("kernel_source_sha") != kernel_source_sha() or not p.get("phases"):
        return None
    return p


def kernel_name(args):
    """The kernel flacenc_hip_encode_stereo_frames dispatches to for this run (qlpc_dispatch.cpp)."""
    if args.block_size == 4096 and args.lpc_order <= 12:
        maxp = 8 if args.lpc_order <= 8 else 10 if args.lpc_order <= 10 else 12
        k = "qlpc_wave4096_kernel<%d,true,true,%s,false>" % (maxp, "true" if args.use_fixed else "false")
        return ("acorr_reference_kernel + " + k) if args.reference_order else k
    return "qlpc_subframe_kernel (+ fixed-LPC batch) + frame_decide_kernel"


def stats(ms):
    import numpy as np
    a = np.asarray(ms, dtype=np.float64)
    return {"min": round(float(a.min()), 4), "median": round(float(np.median(a)), 4),
            "mean": round(float(a.mean()), 4), "max": round(float(a.max()), 4), "n": int(a.size)}


def run(args, world):
    import numpy as np
    import torch
    import torch.distributed as dist

    from flacenc_rs_amd import _capi, shard

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    if args.shared_gpu_test:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ranks_observed = 1
    if world > 1 or args.force_exchange:
        # (--force-exchange with one rank: a real 1-rank RCCL communicator, so that the exchange step of a
        # single-GPU box goes through the same all_gather_into_tensor calls as an 8-GPU run)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.shared_gpu_test:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        one = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(one)  # counted by RCCL itself, not read from the environment
        ranks_observed = int(one.item())
        assert ranks_observed == dist.get_world_size() == world

    n, F, bps = args.block_size, args.frames, args.bps
    # precision 15, Tukey(0.4), max_p 30; candidates Constant / Verbatim / LPC -- the QLPC analysis path
    # the metric names (--use-fixed adds the reference default's fixed-LPC candidate); all stereo
    # assignments allowed
    qcfg = _capi.make_config(lpc_order=args.lpc_order, rice_finest_only=args.finest_rice_order,
                             flags=_capi.FLAG_REFERENCE_SUM_ORDER if args.reference_order else
                             (_capi.FLAG_CANONICAL_SUM_ORDER if args.canonical_order else 0))
    cfg = _capi.make_frame_config(qcfg, use_fixed=args.use_fixed)
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
    residual = torch.empty((F * 2, n), dtype=torch.int32, device=dev)     # the two chosen channels
    handle = _capi.Handle(local_rank)
    exchanging = world > 1 or args.force_exchange
    if exchanging and not args.no_stream_priority:
        # the analysis runs on a high-priority stream: the exchange's small kernels (length derivation, wire packing,
        # the collective's copies, the prefix sum) then take the slots the VALU-bound analysis kernel leaves instead
        # of competing for them (without priorities the forced exchange at one rank cost 12 %)
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    stream = torch.cuda.current_stream()
    comm = torch.cuda.Stream(device=dev)
    payload = exchanging and args.gather == "payload"
    if payload:
        out_stride = handle.frame_bytes_bound(n, bps)
        packed2 = [torch.empty((F, out_stride), dtype=torch.uint8, device=dev) for _ in range(2)]
    wire2 = [torch.empty((F, shard.wire_record_bytes(n)), dtype=torch.uint8, device=dev) for _ in range(2)] if exchanging else None
    capi_comm = None
    if exchanging and args.comm == "capi":
        # the library's own communicator: rank 0's ncclGetUniqueId travels through torch.distributed's object broadcast
        # (any channel the host has would do), every rank joins with its handle, and the gathers below run on `comm`
        uid = [_capi.Handle.comm_unique_id() if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(uid, src=0)
        handle.comm_create(uid[0], rank, world)
        capi_comm = shard.CapiCollective(handle, stream=comm.cuda_stream)
        assert (capi_comm.rank, capi_comm.world) == (rank, world)
    consumed = [None, None]  # event: the exchange that read buffer b has finished
    step_no = [0]
    last_exchange = {}

    def exchange(b):
        # the multi-GPU exchange step (ParSink's ordered gather): byte lengths -> stream order ->
        # stream offsets, then the encoded components themselves in stream order
        # two kernels of the C library + the collectives: records -> wire form + byte lengths in one pass, and the
        # stream offsets straight from the lengths as the all-gather delivers them (rank-major)
        wire = None
        if args.gather in ("records", "payload"):
            wire, _ = shard.records_to_wire_device(handle, results2[b], n, bps, SAMPLE_RATE, rank, world,
                                                   stream=comm.cuda_stream, wire=wire2[b],
                                                   lengths=None if payload else frame_len2[b])
        elif not payload:
            handle.stereo_frame_lengths_device(results2[b].data_ptr(), F, n, bps, SAMPLE_RATE, rank, world,
                                               frame_len2[b].data_ptr(), stream=comm.cuda_stream)
        gathered = shard.all_gather_rank_major(frame_len2[b], world * F, collective=capi_comm)
        lengths_all, offsets, total = shard.stream_offsets_device(handle, gathered, world * F, world,
                                                                  stream=comm.cuda_stream)
        last_exchange.update(lengths_all=lengths_all, offsets=offsets, total=total)
        if wire is not None:
            last_exchange["records_all"] = shard.GatheredRecords(
                shard.all_gather_records(wire, world * F, materialize=False, collective=capi_comm), world * F, n)
        if payload:
            # (run capacity = the packer's bound: no host synchronisation on the exchange stream)
            last_exchange["stream_bytes"] = shard.all_gather_frame_bytes(
                shard.device_place(handle, comm.cuda_stream), packed2[b], frame_len2[b], lengths_all, offsets,
                world * F, run_capacity=F * out_stride, collective=capi_comm)

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
        if payload:
            # Frame::write on the producing GPU: the payload of the exchange
            handle.pack_stereo_frames_device(x.data_ptr(), F, n, n, results2[b].data_ptr(), residual.data_ptr(), n,
                                             bps, SAMPLE_RATE, rank, world, packed2[b].data_ptr(), out_stride,
                                             frame_len2[b].data_ptr(), stream=stream.cuda_stream)
        if exchanging:
            ready = torch.cuda.Event()
            ready.record(stream)
            with torch.cuda.stream(comm):
                comm.wait_event(ready)
                exchange(b)
                consumed[b] = torch.cuda.Event()
                consumed[b].record(comm)

    def fence():
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    # clock spin-up (untimed, before the W warm-up steps): after the idle seconds of set-up the chip needs
    # several milliseconds of load before it holds its clock -- tools/launch_cost.py measures the same kernel
    # at 0.63 ms per 24576 frames in a warm loop against 0.70 right after a short warm-up
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.05:
        handle.encode_stereo_frames_device(cfg, x.data_ptr(), F, n, n, bps, results2[0].data_ptr(),
                                           residual.data_ptr(), n, stream=stream.cuda_stream)
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
    if dist.is_initialized():
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_times = [a.elapsed_time(b) for a, b in ev] if args.steps else [float("nan")]
    kernel_ms = float(np.mean(kernel_times))

    # sanity: nothing in the timed region may have failed
    last_b = (step_no[0] - 1) & 1
    p = np.frombuffer(results2[last_b].cpu().numpy().tobytes(), dtype=_capi.FRAME_RESULT_DTYPE)
    lpc_kind = p["kind"] >= 2
    assert (p["lpc"]["status"][lpc_kind] == 0).all(), "subframe status != 0"
    chosen_bits = int(sum(int(p["bits"][f, r]) for f in range(F) for r in p["role"][f]))
    assign_hist = np.bincount(p["channel_assignment"], minlength=4).tolist()
    exchange_check = None
    if exchanging:
        exchange_check = check_exchange(torch, dist, last_exchange, results2[last_b], frame_len2[last_b], rank,
                                        world, F, packed2[last_b] if payload else None)

    samples_per_step = F * 2 * n  # input channel-samples per rank per step
    value = world * samples_per_step * args.steps / elapsed / 1e6
    if args.shared_gpu_test:
        value = 0.0  # ranks shared a GPU and the collectives went through the host: not a measurement
    achieved = ALGO_BYTES_PER_SAMPLE * samples_per_step / (kernel_ms * 1e-3) / 1e9
    prof = profiled_counters(args)
    phases = profiled_phases(args)
    # share of SIMD cycles with the VALU pipe executing: a ratio of two counters of the profiled runs of this
    # very kernel source (no clock assumed), see tools/make_headline_pmc.py
    valu_frac = round(prof["valu_busy_frac"], 4) if prof and prof.get("valu_busy_frac") else None

    gather_desc = "none"
    if exchanging:
        gather_desc = ("%s all_gather of per-frame byte lengths (4 B/frame) + prefix sum to stream offsets"
                       % ("gloo (shared-GPU functional test)" if args.shared_gpu_test else "RCCL"))
        if args.gather in ("records", "payload"):
            gather_desc += (" + all_gather of the frame records (the encoded SubFrame components; %d of their 752 "
                            "bytes on the wire: Rice-parameter slots beyond the block's finest partition count are "
                            "always zero); every rank holds the whole stream's records, addressed in stream order through a strided view "
                            "of the collective's output" % shard.wire_record_bytes(n))
        if payload:
            gather_desc += " + Frame::write on the producing GPU and all_gather of the packed frame bytes, placed at their stream offsets"
        gather_desc += "; on its own stream, overlapping the next step's analysis"
    # bytes each rank contributes to the collectives of one step (what crosses each of its xGMI links once)
    gather_bytes = None
    if exchanging:
        gather_bytes = {"lengths": 4 * F}
        if args.gather in ("records", "payload"):
            gather_bytes["records_wire"] = F * shard.wire_record_bytes(n)
        if payload:
            # runs travel padded to the packer's bound (run_capacity: nothing is read back to size the collective)
            gather_bytes["payload_padded"] = (F * out_stride + 15) & ~15
            gather_bytes["payload_used"] = int(frame_len2[0].to(torch.int64).sum().item())
        gather_bytes["per_rank_per_step"] = sum(v for k, v in gather_bytes.items() if k != "payload_used")

    out = {
        "metric": "Msamples/s encoded (44.1kHz/16b stereo, block=4096): QLPC analysis path",
        "value": round(value, 2),
        "unit": "Msamples/s",
        "n_gpus": world,
        "ranks_observed": ranks_observed,
        "collective_backend": (dist.get_backend() if dist.is_initialized() else None),
        "exchange_collective": (None if not exchanging else
                                "C ABI: flacenc_hip_comm_create + flacenc_hip_allgather_records_async / flacenc_hip_allgather_async"
                                if capi_comm is not None else "torch.distributed all_gather_into_tensor"),
        **({"shared_gpu_test": True} if args.shared_gpu_test else {}),
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
            "gather": gather_desc,
            "gather_bytes": gather_bytes,
            "exchange_check": exchange_check,
            "subframe_bits_per_sample": round(chosen_bits / (2 * F * n), 4),
            "assignments_indep_left_right_mid": assign_hist,
        },
        "roofline": {
            # what binds the dominant kernel is VALU issue (counters: the pipe executes in `valu_issue_frac` of
            # all SIMD cycles; HBM traffic is 1.04 x the algorithmic bytes at a third of the HBM roof).  achieved /
            # peak / frac stay priced against the HBM roof, the currency SURVEY 8(d) names for this path.
            "bound": "valu_issue",
            "priced_against": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": prof.get("bytes_per_launch") if prof else None,
            "kernel": kernel_name(args),
            "kernel_ms": round(kernel_ms, 4),
            "kernel_ms_stats": stats(kernel_times),
            "algorithmic_bytes_per_launch": ALGO_BYTES_PER_SAMPLE * samples_per_step,
            "valu_issue_frac": valu_frac,
            "valu_insts_per_wave": prof.get("valu_insts_per_wave") if prof else None,
            "counters_from": ("profiles/headline_pmc.json @ kernel source %s" % prof["kernel_source_sha"]) if prof else None,
            # where the instructions go, and the least this formulation can spend (profiles/headline_phases.json)
            "valu_floor_insts_per_wave": phases.get("valu_floor_insts_per_wave") if phases else None,
            "phases": [{"phase": p["phase"], "insts": p["valu_insts_per_wave"], "floor": p["floor"]}
                       for p in phases["phases"]] if phases else None,
        },
    }

    if rank == 0 and world == 1 and not args.no_secondary:
        out["secondary"] = secondary(torch, _capi, handle, args, dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(host, bps, args, n)
        # the order certificate at work (outside the timed region): the device's counters over one more launch of the
        # whole batch, and a sample of the frames against the oracle in the REFERENCE's summation order
        ident = reference_identity(torch, _capi, handle, args, host, x, results2[last_b], residual, cfg, n, bps, F)
        out["config"].update(ident)
    if rank == 0:
        print(json.dumps(out), flush=True)
    handle.close()
    if dist.is_initialized():
        dist.destroy_process_group()


def check_exchange(torch, dist, ex, my_records, my_lengths, rank, world, F, my_packed):
    """After the timed region: what the last exchange delivered must contain this rank's own frames at
    stream positions rank, rank + G, ... and, summed over ranks, account for every frame once."""
    torch.cuda.synchronize()
    lengths_all, offsets = ex["lengths_all"], ex["offsets"]
    ok = bool(torch.equal(lengths_all[rank::world], my_lengths))
    ok &= int(offsets[0]) == 0 and bool((offsets[1:] - offsets[:-1] == lengths_all[:-1].to(offsets.dtype)).all())
    if "records_all" in ex:
        ok &= bool(torch.equal(ex["records_all"].records()[rank::world], my_records))
    if "stream_bytes" in ex:
        sb = ex["stream_bytes"]
        ok &= int(ex["total"]) == int(lengths_all.to(torch.int64).sum())
        for j in (0, F // 2, F - 1):  # spot-check this rank's frames inside the assembled stream
            f = j * world + rank
            o, ln = int(offsets[f]), int(lengths_all[f])
            ok &= bool(torch.equal(sb[o:o + ln], my_packed[j, :ln]))
    total = lengths_all.to(torch.int64).sum().reshape(1).clone()
    if dist.is_initialized():
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=lengths_all.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())
        # every rank must have assembled the same stream
        lo, hi = total.clone(), total.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        ok &= int(lo.item()) == int(hi.item())
    assert ok, "multi-GPU exchange delivered wrong data"
    return {"ok": ok, "stream_frames": int(lengths_all.numel()), "stream_bytes": int(total.item())}


def secondary(torch, _capi, handle, args, dev):
    """Driver-timed side measurements at N = 1 (HIP events on the launch stream, a few launches each):
    the reference's default candidate set (use_fixed), the one-call PCM -> FLAC-frame-bytes path, and a
    tonal low-residual workload (Sine(36,0.4)+Noise(0.04), src/lib.rs:219-221) on the headline kernel."""
    import numpy as np
    n, bps = args.block_size, args.bps
    F = min(args.frames, 24576)  # the side measurements keep round 2's batch (805 MB in, 805 MB out)
    steps, warm = 6, 10  # (10 untimed launches first: after an idle or copy-bound stretch the clock needs ~5 ms to come back)
    stream = torch.cuda.current_stream()
    rec_bytes = _capi.FRAME_RESULT_DTYPE.itemsize
    results = torch.empty((F, rec_bytes), dtype=torch.uint8, device=dev)
    residual = torch.empty((F * 2, n), dtype=torch.int32, device=dev)
    out_stride = handle.frame_bytes_bound(n, bps)
    packed = torch.empty((F, out_stride), dtype=torch.uint8, device=dev)
    lens = torch.zeros(F, dtype=torch.int32, device=dev)
    qcfg = _capi.make_config(lpc_order=args.lpc_order, rice_finest_only=args.finest_rice_order)

    def timed(fn):
        for _ in range(warm):
            fn()
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < 0.03:  # and 30 ms of the same work, so that the timed launches run at the sustained clock
            fn()
            torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for a, b in ev:
            a.record(stream)
            fn()
            b.record(stream)
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in ev]
        return ms

    def entry(ms, extra=None):
        med = float(np.median(ms))
        e = {"frames": F, "ms_per_launch": stats(ms), "Msamples_per_s": round(F * 2 * n / (med * 1e-3) / 1e6, 1),
             "hbm_frac": round(ALGO_BYTES_PER_SAMPLE * F * 2 * n / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if extra:
            e.update(extra)
        return e

    def bits_per_sample():
        p = np.frombuffer(results.cpu().numpy().tobytes(), dtype=_capi.FRAME_RESULT_DTYPE)
        rows = np.arange(F)
        return round(float(sum(int(p["bits"][rows, p["role"][:, c]].sum()) for c in range(2))) / (2 * F * n), 4)

    sec = {}
    noisy = torch.from_numpy(_capi.sigen_frames(F, 2, n, bps, 200.0, 0.4, 0.4, seed=0xF1AC0001)).to(dev)
    for name, use_fixed in (("default_config_use_fixed", True),):
        fcfg = _capi.make_frame_config(qcfg, use_fixed=use_fixed)
        ms = timed(lambda: handle.encode_stereo_frames_device(fcfg, noisy.data_ptr(), F, n, n, bps, results.data_ptr(),
                                                              residual.data_ptr(), n, stream=stream.cuda_stream))
        sec[name] = entry(ms, {"what": "encode_frame decision with the fixed-LPC candidate (reference default "
                                       "SubFrameCoding), same frames", "subframe_bits_per_sample": bits_per_sample()})
    for name, use_fixed in (("pcm_to_frame_bytes", False), ("pcm_to_frame_bytes_use_fixed", True)):
        fcfg = _capi.make_frame_config(qcfg, use_fixed=use_fixed)
        ms = timed(lambda: handle.encode_pack_stereo_frames_device(
            fcfg, noisy.data_ptr(), F, n, n, bps, SAMPLE_RATE, 0, 1, results.data_ptr(), packed.data_ptr(),
            out_stride, lens.data_ptr(), stream=stream.cuda_stream))
        sec[name] = entry(ms, {"what": "one call, PCM in HBM -> FLAC frame bytes in HBM (analysis + decision + "
                                       "Frame::write with both CRCs)",
                               "frame_bytes_per_step": int(lens.to(torch.int64).sum().item())})
    rcfg = _capi.make_frame_config(_capi.make_config(lpc_order=args.lpc_order, rice_finest_only=args.finest_rice_order,
                                                     flags=_capi.FLAG_REFERENCE_SUM_ORDER), use_fixed=False)
    ms = timed(lambda: handle.encode_stereo_frames_device(rcfg, noisy.data_ptr(), F, n, n, bps, results.data_ptr(),
                                                          residual.data_ptr(), n, stream=stream.cuda_stream))
    sec["reference_sum_order"] = entry(ms, {
        "what": "FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER: autocorrelation as one sequential chain per lag "
                "(src/lpc.rs:533-548) by a lane-per-subframe kernel, then the fused kernel without its phase 1; "
                "coefficients bit-identical to the reference's stable build", "subframe_bits_per_sample": bits_per_sample()})
    icfg = _capi.make_frame_config(_capi.make_config(lpc_order=args.lpc_order, rice_finest_only=args.finest_rice_order,
                                                     flags=_capi.FLAG_REFERENCE_SUM_ORDER | _capi.FLAG_INTEGER_PARITY_ONLY),
                                   use_fixed=False)
    ms = timed(lambda: handle.encode_stereo_frames_device(icfg, noisy.data_ptr(), F, n, n, bps, results.data_ptr(),
                                                          residual.data_ptr(), n, stream=stream.cuda_stream))
    sec["reference_sum_order_integer_parity_only"] = entry(ms, {
        "what": "FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER | _INTEGER_PARITY_ONLY (what the Rust / C++ drop-in asks for under a "
                "stable build): the stable build's integers by the order certificate, no second pass on this shape",
        "subframe_bits_per_sample": bits_per_sample()})
    ncfg = _capi.make_frame_config(_capi.make_config(lpc_order=args.lpc_order, rice_finest_only=args.finest_rice_order,
                                                     flags=_capi.FLAG_NIGHTLY_SUM_ORDER), use_fixed=False)
    ms = timed(lambda: handle.encode_stereo_frames_device(ncfg, noisy.data_ptr(), F, n, n, bps, results.data_ptr(),
                                                          residual.data_ptr(), n, stream=stream.cuda_stream))
    sec["nightly_sum_order"] = entry(ms, {
        "what": "FLACENC_HIP_FLAG_NIGHTLY_SUM_ORDER: autocorrelation in the simd-nightly build's order (8 / 16 strided "
                "lane chains per lag + head / foot, src/lpc.rs:439-531) by a kernel with a GPU lane per vector lane; "
                "coefficients bit-identical to the oracle's restatement of that build", "subframe_bits_per_sample": bits_per_sample()})
    # the reference's DEFAULT configuration: order 10, fixed-LPC candidate with ApproxEnt (config.rs defaults)
    d10 = _capi.make_config(lpc_order=10)
    ms = timed(lambda: handle.encode_stereo_frames_device(_capi.make_frame_config(d10, use_fixed=True), noisy.data_ptr(), F, n,
                                                          n, bps, results.data_ptr(), residual.data_ptr(), n,
                                                          stream=stream.cuda_stream))
    sec["reference_default_config_order10"] = entry(ms, {
        "what": "config::Encoder::default(): LPC order 10 + fixed-LPC candidate (ApproxEnt, 16 partitions), decision on the "
                "GPU: qlpc_wave4096_kernel<10,true,true,true,false>", "subframe_bits_per_sample": bits_per_sample()})
    r10 = _capi.make_config(lpc_order=10, flags=_capi.FLAG_REFERENCE_SUM_ORDER)
    ms = timed(lambda: handle.encode_stereo_frames_device(_capi.make_frame_config(r10, use_fixed=True), noisy.data_ptr(), F, n,
                                                          n, bps, results.data_ptr(), residual.data_ptr(), n,
                                                          stream=stream.cuda_stream))
    sec["reference_default_config_order10_reference_sum_order"] = entry(ms, {
        "what": "the same with FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER: two extra passes (autocorrelation chains per lag, "
                "find_sum_abs_f32 chains per estimator partition) in front of the fused kernel -- the stable build's "
                "bytes", "subframe_bits_per_sample": bits_per_sample()})
    # the experimental estimators, on an eighth of the frames: one sequential chain per Gram entry
    Fx = max(F // 8, 256)
    xp = torch.empty((Fx * 4, _capi.PARAMS_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    xr = torch.empty((Fx * 4, n), dtype=torch.int32, device=dev)
    for label, mae in (("direct_mse_order8", 0), ("direct_mse_irls2_order8", 2)):
        xcfg = _capi.make_config(lpc_order=args.lpc_order, window="rectangle", use_direct_mse=True,
                                 mae_optimization_steps=mae)
        ms = timed(lambda: handle.stereo_qlpc_batch_device(xcfg, noisy.data_ptr(), Fx, n, n, bps, xp.data_ptr(),
                                                           xr.data_ptr(), n, stream=stream.cuda_stream))
        med = float(np.median(ms))
        sec[label] = {"frames": Fx, "ms_per_launch": stats(ms), "Msamples_per_s": round(Fx * 2 * n / (med * 1e-3) / 1e6, 1),
                      "what": "flacenc_hip_stereo_qlpc_batch with use_direct_mse (covariance-method LPC, src/lpc.rs:853-903"
                              + (", IRLS with %d re-weighting steps, :814-850" % mae if mae else "") +
                              "), Rectangle window as in report/experimental.config.toml; 4 candidates per frame"}
    del xp, xr
    # what the host-pointer boundary delivers end to end: packed 16-bit interleaved PCM in host memory ->
    # FLAC frame bytes in host memory (flacenc_hip_encode_pcm_stereo: chunked, pinned staging, upload /
    # analysis / download of neighbouring chunks overlapped).  Wall clock, PCIe both ways included.
    host16 = np.ascontiguousarray(noisy.cpu().numpy().transpose(0, 2, 1)).astype("<i2").view(np.uint8).reshape(-1)
    fcfg = _capi.make_frame_config(qcfg, use_fixed=False)
    pcie = {}
    for label, pinned, threads in (("pageable_caller_buffers_1_thread", False, 1), ("pageable_caller_buffers", False, 4),
                                   ("pinned_caller_buffers", True, 4)):
        handle.set_host_threads(threads)
        src = host16
        dst = np.empty(F * (out_stride + 16), np.uint8)
        if pinned:
            src = _capi.pinned_array(host16.size)
            src[:] = host16
            dst = _capi.pinned_array(F * (out_stride + 16))
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            out_bytes, _lens = handle.encode_pcm_stereo(src, fcfg, 2, bps, n, SAMPLE_RATE, out=dst)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        pcie[label] = {"ms": round(best * 1e3, 3), "Msamples_per_s": round(F * 2 * n / best / 1e6, 1),
                       "host_bytes_in": int(host16.size), "host_bytes_out": int(out_bytes.size),
                       "staging_threads": 0 if pinned else threads}
        del src, dst
    sec["pcie_inclusive_pcm_to_frame_bytes"] = dict(
        pcie, what="flacenc_hip_encode_pcm_stereo: host PCM (2 B/sample) -> host frame bytes, best of 3 wall-clock "
                   "runs; never the headline value")
    del noisy
    tonal = torch.from_numpy(_capi.sigen_frames(F, 2, n, bps, 36.0, 0.4, 0.04, seed=0xF1AC0002)).to(dev)
    fcfg = _capi.make_frame_config(qcfg, use_fixed=False)
    ms = timed(lambda: handle.encode_stereo_frames_device(fcfg, tonal.data_ptr(), F, n, n, bps, results.data_ptr(),
                                                          residual.data_ptr(), n, stream=stream.cuda_stream))
    sec["tonal_workload"] = entry(ms, {"what": "headline kernel on sigen Sine(36,0.4)+Noise(0.04) (src/lib.rs:219-221)",
                                       "subframe_bits_per_sample": bits_per_sample()})
    del tonal
    # Real audio: the eight fixture channels the reference's own tests load (tests/golden/testsignal.*.bin, 8192 16-bit
    # samples each; src/test_helper.rs:81-125), cut into stereo frames at a hop of 64 samples and tiled to the batch.
    # Music is strongly coloured: the order certificate recomputes a larger share of it from the
    # reference's chains than of the bench signal -- this row is the headline kernel's rate on such material.
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden")
    names = ("sus109", "sus6", "ras22", "ras103")
    if bps == 16 and n <= 8192 and all(os.path.exists(os.path.join(gold, "testsignal.%s.ch%d.bin" % (nm, c))) for nm in names for c in (0, 1)):
        cut = []
        for nm in names:
            ch = [np.fromfile(os.path.join(gold, "testsignal.%s.ch%d.bin" % (nm, c)), dtype="<i2").astype(np.int32) for c in (0, 1)]
            for t0 in range(0, 8192 - n + 1, 64):
                cut.append(np.stack([ch[0][t0:t0 + n], ch[1][t0:t0 + n]]))
        cut = np.stack(cut)
        real = torch.from_numpy(np.ascontiguousarray(np.tile(cut, ((F + len(cut) - 1) // len(cut), 1, 1))[:F])).to(dev)
        for order, fixed in ((8, False), (10, False), (12, False), (10, True)):
            rcfg = _capi.make_frame_config(_capi.make_config(lpc_order=order), use_fixed=fixed)
            ms = timed(lambda: handle.encode_stereo_frames_device(rcfg, real.data_ptr(), F, n, n, bps, results.data_ptr(),
                                                                  residual.data_ptr(), n, stream=stream.cuda_stream))
            # (the certificate's counters: a hook of the test build, libflacenc_hip_hooks.so -- the same kernel objects; the
            # timed launches above ran on the product library's handle)
            cst = torch.zeros(3, dtype=torch.int32, device=dev)
            hk = hooks_handle(_capi, dev)
            hk.debug_set_cert_stats(cst.data_ptr())
            hk.encode_stereo_frames_device(rcfg, real.data_ptr(), F, n, n, bps, results.data_ptr(), residual.data_ptr(), n,
                                           stream=stream.cuda_stream)
            torch.cuda.synchronize()
            hk.debug_set_cert_stats(0)
            analysed, _tier2, redone = (int(v) for v in cst.cpu().tolist())
            sec["real_audio_fixtures_default_config_order10" if fixed else "real_audio_fixtures_order%d" % order] = entry(ms, {
                "what": "%s on %d distinct stereo frames cut from the reference's real-audio test fixtures, tiled to the batch" % (
                    "the reference's default configuration (order 10, fixed-LPC candidate)" if fixed
                    else "headline kernel (frame decision, no fixed-LPC candidate)", len(cut)),
                "subframe_bits_per_sample": bits_per_sample(),
                "certificate_recomputed_fraction": round(redone / analysed, 4) if analysed else None})
        # the order mode's start-up and small launches.  A fresh handle knows nothing about the material:
        # its first launches run the certified kernel until the counters have a verdict (4096 subframes) -- the row gives the
        # first three launches one by one; and launches of 128 frames = 512 subframes, eight of which make one verdict.
        rcfg8 = _capi.make_frame_config(_capi.make_config(lpc_order=10), use_fixed=False)
        fresh = _capi.Handle(dev.index if dev.index is not None else 0)
        cold = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            fresh.encode_stereo_frames_device(rcfg8, real.data_ptr(), F, n, n, bps, results.data_ptr(), residual.data_ptr(), n,
                                              stream=stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize()
            cold.append(round(e0.elapsed_time(e1), 4))
        sec["real_audio_fixtures_order10_cold_start"] = {
            "frames": F, "ms_first_three_launches": cold,
            "Msamples_per_s_first_launch": round(F * 2 * n / (cold[0] * 1e-3) / 1e6, 1),
            "what": "order 10 on the real-audio frames, first three launches of a fresh handle (scratch growth and the window "
                    "table included in the first; the order mode has no verdict yet: certified kernel)"}
        fs = 128
        ms = timed(lambda: fresh.encode_stereo_frames_device(rcfg8, real.data_ptr(), fs, n, n, bps, results.data_ptr(),
                                                             residual.data_ptr(), n, stream=stream.cuda_stream))
        med = float(np.median(ms))
        sec["real_audio_fixtures_order10_512_subframes_per_launch"] = {
            "frames": fs, "ms_per_launch": stats(ms), "Msamples_per_s": round(fs * 2 * n / (med * 1e-3) / 1e6, 1),
            "what": "order 10 on the real-audio frames in launches of 128 frames = 512 subframes: a sixth of one round of "
                    "workgroups (launch-bound), eight launches to a verdict of the order mode"}
        fresh.close()
        del real
    del results, residual, packed
    # BASELINE configs[2] / [4]: 24-bit stereo blocks of 8192 / 16384 samples at order 24 (the reference's maximum)
    # through the big-block kernels -- all four candidates (L, R, M, S) analysed, f64-fma bound
    for label, bn, bf in (("config3_8192x24bit_order24", 8192, 6144), ("config5_16384x24bit_order24", 16384, 3072)):
        big = torch.from_numpy(_capi.sigen_frames(bf, 2, bn, 24, 200.0, 0.4, 0.1, seed=0xF1AC0003)).to(dev)
        bparams = torch.empty((bf * 4, _capi.PARAMS_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        bres = torch.empty((bf * 4, bn), dtype=torch.int32, device=dev)
        bcfg = _capi.make_config(lpc_order=24)
        ms = timed(lambda: handle.stereo_qlpc_batch_device(bcfg, big.data_ptr(), bf, bn, bn, 24, bparams.data_ptr(),
                                                           bres.data_ptr(), bn, stream=stream.cuda_stream))
        med = float(np.median(ms))
        sec[label] = {"frames": bf, "block_size": bn, "ms_per_launch": stats(ms),
                      "Msamples_per_s": round(bf * 2 * bn / (med * 1e-3) / 1e6, 1),
                      "fp64_fma_frac": round(25 * bf * 4 * bn / (med * 1e-3) / 39.3e12, 4),
                      "what": "flacenc_hip_stereo_qlpc_batch: 4 candidates per frame, 25 f64 fma per analysed sample (the "
                              "stable build's chains on v_mfma_f64_4x4x4: the unflagged order on these shapes) + compute_error on "
                              "v_mfma_i32_16x16x64_i8; fp64_fma_frac = the fma alone against the 78.6 TFLOP/s FP64 vector peak"}
        # BASELINE configs[2] / [4] AS WRITTEN: "LPC order 32" (FLAC's maximum; beyond the reference's own verifier,
        # config.rs:304, hence the extension flag).  33 lags = two 16-lag MFMA blocks + one extra chain.
        bcfg32 = _capi.make_config(lpc_order=32, flags=_capi.FLAG_ALLOW_ORDER_32)
        ms = timed(lambda: handle.stereo_qlpc_batch_device(bcfg32, big.data_ptr(), bf, bn, bn, 24, bparams.data_ptr(),
                                                           bres.data_ptr(), bn, stream=stream.cuda_stream))
        med = float(np.median(ms))
        sec[label.replace("order24", "order32")] = {
            "frames": bf, "block_size": bn, "ms_per_launch": stats(ms),
            "Msamples_per_s": round(bf * 2 * bn / (med * 1e-3) / 1e6, 1),
            "fp64_fma_frac": round(33 * bf * 4 * bn / (med * 1e-3) / 39.3e12, 4),
            "what": "the same call at LPC order 32, the BASELINE config as written (FLACENC_HIP_FLAG_ALLOW_ORDER_32): 33 f64 "
                    "fma per analysed sample in the stable build's order"}
        if bn == 16384:
            # BASELINE configs[4] as written: "experimental config ... block 16384, order 32" = use_direct_mse
            # (src/lpc.rs:853-903, coding.rs:337-347) on 24-bit blocks of 16384 samples, with and without IRLS
            xf = bf // 4
            for xlabel, xorder, mae in (("config5_direct_mse_16384x24bit_order24", 24, 0),
                                        ("config5_direct_mse_16384x24bit_order32", 32, 0),
                                        ("config5_direct_mse_irls2_16384x24bit_order24", 24, 2)):
                xcfg = _capi.make_config(lpc_order=xorder, window="rectangle", use_direct_mse=True, mae_optimization_steps=mae,
                                         flags=_capi.FLAG_ALLOW_ORDER_32 if xorder > 24 else 0)
                ms = timed(lambda: handle.stereo_qlpc_batch_device(xcfg, big.data_ptr(), xf, bn, bn, 24, bparams.data_ptr(),
                                                                   bres.data_ptr(), bn, stream=stream.cuda_stream))
                med = float(np.median(ms))
                sec[xlabel] = {"frames": xf, "block_size": bn, "ms_per_launch": stats(ms),
                               "Msamples_per_s": round(xf * 2 * bn / (med * 1e-3) / 1e6, 1),
                               "what": "flacenc_hip_stereo_qlpc_batch with use_direct_mse at order %d%s, Rectangle window "
                                       "(report/experimental.config.toml); 4 candidates per frame" %
                                       (xorder, ", 2 IRLS steps" if mae else "")}
        if bn == 8192:
            # the same shape in the stable build's own summation order (lane-per-subframe autocorrelation chains in
            # front of the Levinson batch and the residual kernel)
            rcfg24 = _capi.make_config(lpc_order=24, flags=_capi.FLAG_REFERENCE_SUM_ORDER)
            ms = timed(lambda: handle.stereo_qlpc_batch_device(rcfg24, big.data_ptr(), bf, bn, bn, 24, bparams.data_ptr(),
                                                               bres.data_ptr(), bn, stream=stream.cuda_stream))
            med = float(np.median(ms))
            sec["config3_8192x24bit_order24_reference_sum_order"] = {
                "frames": bf, "block_size": bn, "ms_per_launch": stats(ms),
                "Msamples_per_s": round(bf * 2 * bn / (med * 1e-3) / 1e6, 1),
                "what": "flacenc_hip_stereo_qlpc_batch with FLACENC_HIP_FLAG_REFERENCE_SUM_ORDER"}
            # the same blocks at the reference's default order 10 (big-block kernels since round 3's end; the
            # generic kernel before: 79 G samples/s)
            ocfg10 = _capi.make_config(lpc_order=10)
            ms = timed(lambda: handle.stereo_qlpc_batch_device(ocfg10, big.data_ptr(), bf, bn, bn, 24, bparams.data_ptr(),
                                                               bres.data_ptr(), bn, stream=stream.cuda_stream))
            med = float(np.median(ms))
            sec["blocks_8192x24bit_order10"] = {
                "frames": bf, "block_size": bn, "ms_per_launch": stats(ms),
                "Msamples_per_s": round(bf * 2 * bn / (med * 1e-3) / 1e6, 1),
                "what": "flacenc_hip_stereo_qlpc_batch at the default LPC order on 8192-sample 24-bit blocks"}
            # the same shape through the frame-level call with the reference's default candidate set (fixed-LPC
            # on): QLPC + fixed candidates for L, R, M, S, the frame decision, then Frame::write
            bfr = torch.empty((bf, rec_bytes), dtype=torch.uint8, device=dev)
            bch = torch.empty((bf * 2, bn), dtype=torch.int32, device=dev)
            bstride = (handle.frame_bytes_bound(bn, 24) + 15) // 16 * 16
            bout = torch.empty((bf, bstride), dtype=torch.uint8, device=dev)
            blen = torch.zeros(bf, dtype=torch.int32, device=dev)
            dcfg = _capi.make_frame_config(bcfg, use_fixed=True)

            def frames_default():
                handle.encode_stereo_frames_device(dcfg, big.data_ptr(), bf, bn, bn, 24, bfr.data_ptr(), bch.data_ptr(), bn,
                                                   stream=stream.cuda_stream)
                handle.pack_stereo_frames_device(big.data_ptr(), bf, bn, bn, bfr.data_ptr(), bch.data_ptr(), bn, 24, 96000, 0, 1,
                                                 bout.data_ptr(), bstride, blen.data_ptr(), stream=stream.cuda_stream)
            ms = timed(frames_default)
            med = float(np.median(ms))
            sec["config3_frames_default_config"] = {
                "frames": bf, "block_size": bn, "ms_per_launch": stats(ms),
                "Msamples_per_s": round(bf * 2 * bn / (med * 1e-3) / 1e6, 1),
                "what": "flacenc_hip_encode_stereo_frames (order 24, fixed-LPC candidate on: 8 candidate analyses per frame, "
                        "decision) + flacenc_hip_pack_stereo_frames: PCM in HBM -> FLAC frame bytes in HBM"}
            del bfr, bch, bout, blen
        del big, bparams, bres
    # BASELINE configs[3]: 44.1 kHz / 16-bit 8-channel frames of 4096 samples at the default order 10 -- eight independent
    # subframes per frame (encode_frame_impl(Independent(8)), coding.rs:537-541): the candidates alone, then the frame-level
    # pipeline with the reference's default candidate set (encode_frames + pack_frames: PCM in HBM -> frame bytes in HBM)
    c8f, c8n, c8ch = 6144, 4096, 8
    x8 = torch.from_numpy(_capi.sigen_frames(c8f, c8ch, c8n, 16, 200.0, 0.4, 0.1, seed=0xF1AC0004)).to(dev)
    p8 = torch.empty((c8f * c8ch, _capi.PARAMS_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    r8 = torch.empty((c8f * c8ch, c8n), dtype=torch.int32, device=dev)
    b8 = torch.full((c8f * c8ch,), 16, dtype=torch.uint8, device=dev)
    cfg8 = _capi.make_config(lpc_order=10)
    ms = timed(lambda: handle.qlpc_batch_device(cfg8, x8.data_ptr(), c8f * c8ch, c8n, c8n, b8.data_ptr(), p8.data_ptr(),
                                                r8.data_ptr(), c8n, stream=stream.cuda_stream))
    med = float(np.median(ms))
    sec["config4_8ch_4096x16bit_order10"] = {
        "frames": c8f, "channels": c8ch, "block_size": c8n, "ms_per_launch": stats(ms),
        "Msamples_per_s": round(c8f * c8ch * c8n / (med * 1e-3) / 1e6, 1),
        "hbm_frac": round(ALGO_BYTES_PER_SAMPLE * c8f * c8ch * c8n / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "what": "flacenc_hip_qlpc_batch on 8-channel frames: 8 independent estimated_qlpc per frame, every residual row written"}
    res8 = torch.empty((c8f * c8ch, _capi.CHANNEL_RESULT_DTYPE.itemsize), dtype=torch.uint8, device=dev)
    stride8 = (handle.frame_bytes_bound_channels(c8ch, c8n, 16) + 15) // 16 * 16
    out8 = torch.empty((c8f, stride8), dtype=torch.uint8, device=dev)
    len8 = torch.zeros(c8f, dtype=torch.int32, device=dev)
    fcfg8 = _capi.make_frame_config(cfg8, use_fixed=True)

    def frames8():
        handle.encode_frames_device(fcfg8, x8.data_ptr(), c8f, c8ch, c8n, c8n, 16, res8.data_ptr(), r8.data_ptr(), c8n,
                                    stream=stream.cuda_stream)
        handle.pack_frames_device(x8.data_ptr(), c8f, c8ch, c8n, c8n, res8.data_ptr(), r8.data_ptr(), c8n, 16, SAMPLE_RATE, 0, 1,
                                  out8.data_ptr(), stride8, len8.data_ptr(), stream=stream.cuda_stream)
    ms = timed(frames8)
    med = float(np.median(ms))
    sec["config4_8ch_frames_default_config"] = {
        "frames": c8f, "channels": c8ch, "block_size": c8n, "ms_per_launch": stats(ms),
        "Msamples_per_s": round(c8f * c8ch * c8n / (med * 1e-3) / 1e6, 1),
        "frame_bytes_per_step": int(len8.to(torch.int64).sum().item()),
        "what": "flacenc_hip_encode_frames (order 10, fixed-LPC candidate on, encode_subframe's choice per channel) + "
                "flacenc_hip_pack_frames: 8-channel PCM in HBM -> FLAC frame bytes in HBM"}
    del x8, p8, r8, b8, res8, out8, len8
    # blocks below a wave's worth of Rice partitions (qlpc_subwave_kernel: 4 subframes of 1152 samples per wave): the
    # four candidates of every frame, and the frame-level call with the reference's default candidate set in one launch
    sn, sfr = 1152, 16384
    shost = _capi.sigen_frames(sfr, 2, sn, 16, 200.0, 0.4, 0.1, seed=0xF1AC0003)
    small = torch.from_numpy(shost).to(dev)
    sparams = torch.empty((sfr * 4, 352), dtype=torch.uint8, device=dev)
    sres = torch.empty((sfr * 4, sn), dtype=torch.int32, device=dev)
    scfg = _capi.make_config(lpc_order=8)
    ms = timed(lambda: handle.stereo_qlpc_batch_device(scfg, small.data_ptr(), sfr, sn, sn, 16, sparams.data_ptr(),
                                                       sres.data_ptr(), sn, stream=stream.cuda_stream))
    med = float(np.median(ms))
    sec["blocks_1152x16bit_order8"] = {
        "frames": sfr, "block_size": sn, "ms_per_launch": stats(ms),
        "Msamples_per_s": round(sfr * 2 * sn / (med * 1e-3) / 1e6, 1),
        "what": "flacenc_hip_stereo_qlpc_batch on 1152-sample blocks (16 Rice partitions of 72): the sub-wave kernel, "
                "4 subframes per wave (the generic kernel until round 4: 98 G samples/s); since round 6 behind a pass of "
                "the reference's chains for every subframe (acorr_reference_mfma_kernel): the reference's integers and R[] on "
                "100 % of the subframes, the same cost on every material (round 5, chunk tree, T2: 203 G samples/s)"}
    sfres = torch.empty((sfr, rec_bytes), dtype=torch.uint8, device=dev)
    sfcfg = _capi.make_frame_config(scfg, use_fixed=True)
    ms = timed(lambda: handle.encode_stereo_frames_device(sfcfg, small.data_ptr(), sfr, sn, sn, 16, sfres.data_ptr(),
                                                          sres.data_ptr(), sn, stream=stream.cuda_stream))
    med = float(np.median(ms))
    sec["frames_1152x16bit_default_candidates"] = {
        "frames": sfr, "block_size": sn, "ms_per_launch": stats(ms),
        "Msamples_per_s": round(sfr * 2 * sn / (med * 1e-3) / 1e6, 1),
        "what": "flacenc_hip_encode_stereo_frames_async on 1152-sample blocks, QLPC + fixed-LPC candidates of L, R, M, S, "
                "encode_frame's decision and the two chosen rows in ONE launch of the sub-wave kernel (three launches of "
                "the generic kernels until round 4: 54 G samples/s)"}
    del small, sparams, sres, sfres
    return sec


# ---------------------------------------------------------------------------------------------
def dry_run(args, world):
    """CPU check of what cannot be exercised on a 1-GPU box: that `--gpus N` really yields N ranks and
    that sharding + the ordered exchange deliver every frame once, in stream order, on every rank.
    The analysis is replaced by stand-in records (frame number stamped into the record, a length that
    is a function of the frame number); the collectives and shard.py are the real ones.  Measures
    nothing: `value` is 0 and `dry_run` is true."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from flacenc_rs_amd import shard

    rank = int(os.environ.get("RANK", "0"))
    ranks_observed = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend)
        one = torch.ones(1, dtype=torch.int32)
        dist.all_reduce(one)
        ranks_observed = int(one.item())
        assert ranks_observed == dist.get_world_size() == world
    F = min(args.frames, 64)
    total_frames = world * F
    rec_bytes = 752

    def length_of(f):
        return 1000 + (f * 7919) % 977

    mine = list(shard.frames_of_rank(total_frames, rank, world))
    recs = np.zeros((F, rec_bytes), np.uint8)
    for j, f in enumerate(mine):
        recs[j, :4] = np.frombuffer(np.uint32(f).tobytes(), np.uint8)
        recs[j, 4:] = (f * 31 + np.arange(rec_bytes - 4)) & 0xFF
    lens = torch.tensor([length_of(f) for f in mine], dtype=torch.int32)
    cap = 2048
    packed = torch.zeros((F, cap), dtype=torch.uint8)
    for j, f in enumerate(mine):
        packed[j, :length_of(f)] = torch.arange(length_of(f), dtype=torch.int64).add(f).remainder(251).to(torch.uint8)

    def host_place(src, src_offsets, lengths, dst, dst_offsets):
        # dry-run stand-in for flacenc_hip_place_frames_async (the product passes shard.device_place)
        s, d = src.reshape(-1), dst.reshape(-1)
        for so, ln, do in zip(src_offsets.tolist(), lengths.tolist(), dst_offsets.tolist()):
            d[do:do + ln] = s[so:so + ln]

    ok = True
    for _ in range(max(1, args.steps)):
        # the sequence of the real exchange step: the collective's rank-major output, then what
        # flacenc_hip_stream_offsets_async does with it (host statement) -- and the older re-ordering path beside it
        gathered_rm = shard.all_gather_rank_major(lens, total_frames)
        lengths_all, offsets, total = shard.stream_offsets_from_rank_major(gathered_rm, total_frames, world)
        lengths_old = shard.all_gather_frame_lengths(lens, total_frames)
        ok &= bool(torch.equal(lengths_all, lengths_old))
        want = [length_of(f) for f in range(total_frames)]
        ok &= lengths_all.tolist() == want
        ok &= offsets.tolist() == np.concatenate([[0], np.cumsum(want)[:-1]]).tolist() and int(total) == sum(want)
        if args.gather in ("records", "payload"):
            # (stand-in records are dense, so they go as they are; the wire format of real records is
            # exercised separately below)
            ordered = shard.all_gather_records(torch.from_numpy(recs), total_frames).numpy()
            sparse = torch.from_numpy(recs.copy())
            sparse[:, 48 + 96 + 64:48 + 352] = 0
            sparse[:, 48 + 352 + 96 + 64:] = 0
            gathered = shard.all_gather_frame_records(sparse, total_frames, 4096)
            back = gathered.records()
            ok &= all(bool(torch.equal(gathered.frame_wire(f), shard.records_to_wire(back[f:f + 1], 4096)[0]))
                      for f in (0, total_frames // 2, total_frames - 1))
            mine_back = back[rank::world][:F]
            ok &= bool(torch.equal(mine_back, sparse)) and back.shape == (total_frames, 752)
            ids = ordered[:, :4].copy().view(np.uint32).reshape(-1)
            ok &= ids.tolist() == list(range(total_frames))
            ok &= all(np.array_equal(ordered[f, 4:], (f * 31 + np.arange(rec_bytes - 4)) & 0xFF)
                      for f in range(total_frames))
        if args.gather == "payload":
            sb = shard.all_gather_frame_bytes(host_place, packed, lens, lengths_all, offsets, total_frames)
            ok &= sb.numel() >= sum(want)
            for f in range(total_frames):
                o = int(offsets[f])
                w = torch.arange(want[f], dtype=torch.int64).add(f).remainder(251).to(torch.uint8)
                ok &= bool(torch.equal(sb[o:o + want[f]], w))
    if world > 1:
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())
    if rank == 0:
        print(json.dumps({
            "metric": "Msamples/s encoded (44.1kHz/16b stereo, block=4096): QLPC analysis path",
            "dry_run": True, "value": 0.0, "unit": "Msamples/s", "n_gpus": world,
            "ranks_observed": ranks_observed, "backend": args.backend, "steps": args.steps, "warmup": args.warmup,
            "config": {"workload": "dry run: stand-in records, real sharding + collectives; nothing measured",
                       "gather": args.gather, "exchange_check": {"ok": ok, "stream_frames": total_frames}},
        }), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if not ok:
        sys.exit(1)


# ---------------------------------------------------------------------------------------------
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


_HOOKS_HANDLE = {}


def hooks_handle(_capi, dev):
    """One handle of the hooks build per device, made on first use and never inside a timed region."""
    idx = dev.index if hasattr(dev, "index") and dev.index is not None else 0
    if idx not in _HOOKS_HANDLE:
        _HOOKS_HANDLE[idx] = _capi.Handle(idx, hooks=True)
    return _HOOKS_HANDLE[idx]


def reference_identity(torch, _capi, handle, args, host, x, results, residual, cfg, n, bps, F):
    """How the default mode relates to the reference's stable build (src/lpc.rs:533-548), measured after the timed region:
    `reference_identical_fraction` = frames of a sample whose decision record and two residual rows equal the oracle's in
    ACORR_REFERENCE order (1.0 by construction where the order is certified); `certificate` = the kernel's own counters
    over one launch of the whole batch.  Part of the cpu_baseline leg: the oracle is the checker here, never the product."""
    import numpy as np

    from oracle import oracle as orc

    stats = torch.zeros(3, dtype=torch.int32, device=x.device)
    hk = hooks_handle(_capi, x.device)  # (libflacenc_hip_hooks.so: the product's kernel objects + the counters' hook)
    hk.debug_set_cert_stats(stats.data_ptr())
    hk.encode_stereo_frames_device(cfg, x.data_ptr(), F, n, n, bps, results.data_ptr(), residual.data_ptr(), n,
                                   stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    hk.debug_set_cert_stats(0)
    analysed, tier2, redone = (int(v) for v in stats.cpu().tolist())
    K = min(F, 384)
    g = np.frombuffer(results[:K].cpu().numpy().tobytes(), dtype=_capi.FRAME_RESULT_DTYPE)
    grows = residual[:2 * K].cpu().numpy().reshape(K, 2, n)
    flags = (_capi.FLAG_REFERENCE_SUM_ORDER if args.reference_order else 0)
    ocfg = orc.make_config(lpc_order=args.lpc_order, acorr=orc.ACORR_REFERENCE, rice_finest_only=args.finest_rice_order)
    ofc = orc.make_frame_config(ocfg, use_fixed=args.use_fixed)
    want, wrows = orc.encode_stereo_frames_cfg(host[:K], bps, ofc)
    same = np.array([bool(g[f] == want[f]) and np.array_equal(grows[f], wrows[f]) for f in range(K)])
    return {
        "reference_identical_fraction": round(float(same.mean()), 6),
        "reference_identity": {
            "what": "decision record (752 B) + the two chosen residual rows of each sampled frame against the oracle in the "
                    "stable build's summation order (ACORR_REFERENCE)",
            "frames_compared": int(K), "frames_identical": int(same.sum()),
            "certificate": {"subframes_analysed": analysed, "needed_rows_of_inverse": tier2, "recomputed_from_reference_chains": redone,
                            "recomputed_fraction": round(redone / analysed, 6) if analysed else None,
                            "flags": flags},
        },
    }


def cpu_baseline(host, bps, args, n):
    """The oracle (a port of the reference's path in reference summation order) timed on this
    box's host cores over a bounded sample of the same frames."""
    import numpy as np

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
        "build": "oracle/Makefile: cc -O3 -std=gnu99 -march=x86-64-v3 -ffp-contract=off -fno-fast-math (explicit fma() where the "
                 "reference calls mul_add)",
        "sample": f"{sample_frames} of the same stereo frames x {repeats} passes, {cores} threads "
                  "(frame-parallel pool like src/par.rs), 4 analyses per frame",
    }


if __name__ == "__main__":
    main()
