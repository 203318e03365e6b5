"""Frame sharding across the GPUs of one node and the ordered gather of the results.

The reference's only parallelism is a frame-level worker pool (src/par.rs): frames go to
whichever worker is free and `ParSink` (src/par.rs:67-95) re-orders finished frames by
frame number.  Here the pool is the node's GPUs, one process per GPU: frame f belongs to
rank f mod G (round-robin, BASELINE config 4), every rank analyses its frames with no
communication.  What the ordered gather has to exchange is only what places a frame in the
output stream: its byte length (4 B per frame, flacenc_hip_stereo_frame_lengths) -- one small
all-gather over RCCL/xGMI followed by a prefix sum gives every rank the stream offset of each
of its frames; residuals, parameter records and packed frame bytes stay on the GPU that
produced them and leave it by D2H at those offsets.  (`all_gather_records` moves whole
fixed-size records instead, for callers that want every decision on every rank.)
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def frames_of_rank(n_frames_total: int, rank: int, world: int) -> range:
    """Stream frame numbers owned by `rank`: f = rank, rank + G, rank + 2G, ..."""
    return range(rank, n_frames_total, world)


def local_frame_count(n_frames_total: int, rank: int, world: int) -> int:
    return len(frames_of_rank(n_frames_total, rank, world))


class CapiCollective:
    """The ordered gather's collective through the C ABI instead of torch.distributed: the RCCL communicator a handle owns
    (flacenc_hip_comm_create) and flacenc_hip_allgather_records_async / flacenc_hip_allgather_async on `stream` -- what a
    Rust or C++ host with one process per GPU calls (src/par.rs:67-95 across processes).  Passed as `collective=` to the
    gathers below; rank and world are the communicator's."""

    def __init__(self, handle, stream: int | None = None):
        self.handle = handle
        self.stream = stream
        self.rank, self.world = handle.comm_info()
        assert self.world >= 1, "flacenc_hip_comm_create first"

    def records(self, gathered: torch.Tensor, local: torch.Tensor, n_local: int, n_total: int):
        """rank-major, ranks one frame short zero-padded by the library (no padded copy on this side)"""
        rec = local.element_size() * (local.numel() // max(local.shape[0], 1)) if local.shape[0] else gathered.element_size() * (gathered.numel() // gathered.shape[0])
        assert local.is_contiguous() and gathered.is_contiguous() and local.is_cuda and gathered.is_cuda
        self.handle.allgather_records_device(local.data_ptr() if n_local else 0, n_local, n_total, rec, gathered.data_ptr(), stream=self.stream)

    def plain(self, gathered: torch.Tensor, local: torch.Tensor):
        assert local.is_contiguous() and gathered.is_contiguous() and gathered.numel() * gathered.element_size() == self.world * local.numel() * local.element_size()
        self.handle.allgather_device(local.data_ptr(), gathered.data_ptr(), local.numel() * local.element_size(), stream=self.stream)


def _world_rank(group, collective):
    if collective is not None:
        return collective.world, collective.rank
    return dist.get_world_size(group), dist.get_rank(group)


def all_gather_records(local: torch.Tensor, n_frames_total: int, group=None, materialize: bool = True, collective=None) -> torch.Tensor:
    """All-gather per-frame records and return them in stream (frame-number) order.
    `materialize=False` returns the stream order as a strided view [ceil(F / G), G, ...] of the collective's
    output (frame f at [f // G, f % G]) instead of a re-ordered copy -- nothing but the collective runs.

    `local` is [n_local_frames, ...] (any trailing shape / dtype) holding this rank's frames in
    the order of `frames_of_rank`.  Ranks may own different frame counts (n_frames_total not a
    multiple of the world size); shorter ranks are padded for the collective.  The result is
    [n_frames_total, ...] on every rank -- what ParSink::finalize hands to the stream writer.
    """
    if collective is None and (not dist.is_available() or not dist.is_initialized()):
        assert local.shape[0] == n_frames_total
        return local if materialize else local.unsqueeze(1)
    world, rank = _world_rank(group, collective)
    per_rank = (n_frames_total + world - 1) // world
    n_local = local_frame_count(n_frames_total, rank, world)
    assert local.shape[0] == n_local, (local.shape, n_local)
    gathered = torch.empty((world * per_rank,) + tuple(local.shape[1:]), dtype=local.dtype,
                           device=local.device)
    if collective is not None:
        collective.records(gathered, local.contiguous(), n_local, n_frames_total)
    else:
        if n_local < per_rank:
            pad = torch.zeros((per_rank - n_local,) + tuple(local.shape[1:]), dtype=local.dtype,
                              device=local.device)
            local = torch.cat([local, pad], dim=0)
        local = local.contiguous()
        dist.all_gather_into_tensor(gathered, local, group=group)
    # gathered[r * per_rank + j] is stream frame j * world + r  ->  transpose (r, j) -> (j, r)
    g = gathered.view((world, per_rank) + tuple(local.shape[1:]))
    if not materialize:
        return g.transpose(0, 1)  # [per_rank, world, ...]: frame f at [f // world, f % world], no copy
    ordered = g.transpose(0, 1).reshape((world * per_rank,) + tuple(local.shape[1:]))
    return ordered[:n_frames_total]


# ---- wire format of the 752-byte stereo frame records -----------------------------------------------
# flacenc_hip_stereo_frame_result = 48 bytes of frame fields + two 352-byte subframe records, each ending in
# rice_params[256].  A block of n samples has at most 2^finest_order(n) partitions (64 for n = 4096), the
# rest of that array is always zero: the exchange moves the records without those tails and restores them.
_FRAME_HEAD = 48
_SUB_BYTES = 352
_SUB_FIXED = 96  # coefs[32] i16 + order/shift/precision/rice_order + status + code_bits/subframe_bits/sum_quotients


def finest_partitions(block_size: int) -> int:
    """2^finest_partition_order (src/rice.rs:157-165) for a warm-up of at most 64 samples."""
    order, n = 0, block_size
    while order < 8 and n % 2 == 0 and n // 2 >= 64:
        n //= 2
        order += 1
    return 1 << order


def wire_record_bytes(block_size: int) -> int:
    return _FRAME_HEAD + 2 * (_SUB_FIXED + finest_partitions(block_size))


def records_to_wire(records: torch.Tensor, block_size: int) -> torch.Tensor:
    """[F, 752] uint8 -> [F, wire_record_bytes] uint8 (drops the always-zero tails of rice_params)."""
    parts = finest_partitions(block_size)
    a = _FRAME_HEAD
    b = a + _SUB_BYTES
    return torch.cat([records[:, :a + _SUB_FIXED + parts], records[:, b:b + _SUB_FIXED + parts]], dim=1).contiguous()


def records_from_wire(wire: torch.Tensor, block_size: int) -> torch.Tensor:
    parts = finest_partitions(block_size)
    keep = _SUB_FIXED + parts
    out = torch.zeros((wire.shape[0], _FRAME_HEAD + 2 * _SUB_BYTES), dtype=torch.uint8, device=wire.device)
    out[:, :_FRAME_HEAD + keep] = wire[:, :_FRAME_HEAD + keep]
    out[:, _FRAME_HEAD + _SUB_BYTES:_FRAME_HEAD + _SUB_BYTES + keep] = wire[:, _FRAME_HEAD + keep:]
    return out


class GatheredRecords:
    """What all_gather_frame_records delivers on every rank: the whole stream's frame records in wire format,
    in stream order, as a strided view of the collective's output (no re-ordering or expansion pass: a stream
    writer reads frame f at wire[f // G, f % G]).  records() materialises the full-size 752-byte records."""

    def __init__(self, wire: torch.Tensor, n_frames_total: int, block_size: int):
        self.wire, self.n_frames_total, self.block_size = wire, n_frames_total, block_size

    def frame_wire(self, f: int) -> torch.Tensor:
        return self.wire[f // self.wire.shape[1], f % self.wire.shape[1]]

    def records(self) -> torch.Tensor:
        flat = self.wire.reshape(-1, self.wire.shape[-1])[: self.n_frames_total]
        return records_from_wire(flat, self.block_size)


def all_gather_frame_records(records: torch.Tensor, n_frames_total: int, block_size: int, group=None) -> GatheredRecords:
    """All-gather the stereo frame records (the encoded SubFrame components) in wire format; every rank ends
    up with all of them, addressable in stream order (GatheredRecords)."""
    wire = all_gather_records(records_to_wire(records, block_size), n_frames_total, group=group, materialize=False)
    return GatheredRecords(wire, n_frames_total, block_size)


def all_gather_rank_major(local: torch.Tensor, n_frames_total: int, group=None, collective=None) -> torch.Tensor:
    """The collective alone: [world * ceil(F / G), ...] as all_gather_into_tensor delivers it (row r * per_rank + j =
    stream frame j * G + r; shorter ranks zero-padded).  flacenc_hip_stream_offsets_async reads this layout."""
    if collective is None and (not dist.is_available() or not dist.is_initialized()):
        assert local.shape[0] == n_frames_total
        return local.contiguous()
    world, rank = _world_rank(group, collective)
    per_rank = (n_frames_total + world - 1) // world
    n_local = local_frame_count(n_frames_total, rank, world)
    assert local.shape[0] == n_local, (local.shape, n_local)
    gathered = torch.empty((world * per_rank,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    if collective is not None:
        collective.records(gathered, local.contiguous(), n_local, n_frames_total)
        return gathered
    if n_local < per_rank:
        pad = torch.zeros((per_rank - n_local,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], dim=0)
    dist.all_gather_into_tensor(gathered, local.contiguous(), group=group)
    return gathered


def records_to_wire_device(handle, records: torch.Tensor, block_size: int, bits_per_sample: int, sample_rate: int,
                           first_frame_number: int, frame_number_step: int, stream: int | None = None,
                           wire: torch.Tensor | None = None, lengths: torch.Tensor | None = None):
    """records_to_wire + the frames' byte lengths on the GPU, one kernel (flacenc_hip_stereo_frame_wire_async).
    Returns (wire [F, wire_record_bytes] uint8, lengths [F] int32); both can be passed in to be reused."""
    assert records.is_cuda and records.dtype == torch.uint8 and records.is_contiguous(), "device records (no host fallback)"
    n = records.shape[0]
    wb = wire_record_bytes(block_size)
    assert handle.frame_wire_bytes(block_size) == wb
    if wire is None:
        wire = torch.empty((n, wb), dtype=torch.uint8, device=records.device)
    if lengths is None:
        lengths = torch.empty(n, dtype=torch.int32, device=records.device)
    assert wire.shape == (n, wb) and wire.is_contiguous() and lengths.shape == (n,) and lengths.dtype == torch.int32
    handle.stereo_frame_wire_device(records.data_ptr(), n, block_size, bits_per_sample, sample_rate, first_frame_number,
                                    frame_number_step, wire.data_ptr(), wb, lengths.data_ptr(), stream=stream)
    return wire, lengths


def stream_offsets_device(handle, gathered_lengths: torch.Tensor, n_frames_total: int, world: int,
                          header_bytes: int = 0, stream: int | None = None):
    """stream_offsets on the GPU straight from all_gather_rank_major's output (int32 lengths): one kernel
    (flacenc_hip_stream_offsets_async) instead of the re-ordering copy + widening + scan + two elementwise passes.
    Returns (lengths_all int32 [F] in stream order, offsets int64 [F], total 0-d int64)."""
    assert gathered_lengths.is_cuda and gathered_lengths.dtype == torch.int32 and gathered_lengths.is_contiguous()
    dev = gathered_lengths.device
    lengths_all = torch.empty(n_frames_total, dtype=torch.int32, device=dev)
    offsets = torch.empty(n_frames_total, dtype=torch.int64, device=dev)
    total = torch.empty((), dtype=torch.int64, device=dev)
    handle.stream_offsets_device(gathered_lengths.data_ptr(), n_frames_total, world, header_bytes,
                                 lengths_all.data_ptr(), offsets.data_ptr(), total.data_ptr(), stream=stream)
    return lengths_all, offsets, total


def stream_offsets_from_rank_major(gathered_lengths: torch.Tensor, n_frames_total: int, world: int, header_bytes: int = 0):
    """Host statement (torch ops, any device) of flacenc_hip_stream_offsets_async: all_gather_rank_major's lengths
    [world * ceil(F / G)] -> (lengths in stream order [F], offsets int64 [F], total).  Frame f = j * G + r sits at
    row r * per_rank + j.  The dry run and the CPU tests check the layout contract with it; on the GPU the kernel runs."""
    per_rank = (n_frames_total + world - 1) // world
    assert gathered_lengths.numel() == world * per_rank
    lengths = gathered_lengths.view(world, per_rank).transpose(0, 1).reshape(-1)[:n_frames_total].contiguous()
    offsets, total = stream_offsets(lengths, header_bytes)
    return lengths, offsets, total


def all_gather_frame_lengths(local_lengths: torch.Tensor, n_frames_total: int, group=None) -> torch.Tensor:
    """Byte lengths of all frames in stream order, on every rank (ParSink's ordering, src/par.rs:67-95,
    reduced to what it needs).  `local_lengths` is this rank's [n_local_frames] integer tensor."""
    return all_gather_records(local_lengths, n_frames_total, group=group)


def all_gather_frame_bytes(place, packed: torch.Tensor, local_lengths: torch.Tensor, lengths_all: torch.Tensor,
                           offsets: torch.Tensor, n_frames_total: int, group=None,
                           run_capacity: int | None = None, collective=None) -> torch.Tensor:
    """All-gather the packed FLAC frames themselves and return the assembled frame stream (uint8, frames
    back to back in frame-number order) on every rank -- ParSink::finalize's output (src/par.rs:82-94).

    `packed` is this rank's pack output [n_local_frames, out_stride] (frame j in row j, `local_lengths[j]`
    bytes used); `lengths_all` / `offsets` are all_gather_frame_lengths / stream_offsets of the whole
    stream.  Frames are variable-length, so: (1) this rank's rows are compacted into one contiguous run,
    (2) runs are all-gathered padded to the longest rank's run, (3) every frame is copied from its place
    in its producer's run to its stream offset.  Steps 1 and 3 are `place(src, src_offsets, lengths, dst,
    dst_offsets)` = flacenc_hip_place_frames_async on the GPU (Handle.place_frames_device wrapped by the
    caller); there is no host fallback here.  By default the runs are padded to the longest run actually
    produced, which costs two host synchronisations (that size, and the stream's total); with `run_capacity`
    -- an upper bound on any rank's run, e.g. n_local * out_stride -- nothing is read back: runs travel
    padded to that bound and the result is a buffer of min(world * capacity, n_frames_total * out_stride) bytes whose
    first `stream_offsets(...)[1]` bytes are the stream -- callers slice with that total (the rest is uninitialised)."""
    if collective is not None:
        world, rank = collective.world, collective.rank
    else:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    per_rank = (n_frames_total + world - 1) // world
    n_local = local_frame_count(n_frames_total, rank, world)
    assert packed.shape[0] == n_local and packed.dtype == torch.uint8 and packed.is_contiguous()
    dev = packed.device
    # lengths by (position in the rank's run j, rank r); absent frames of shorter ranks have length 0
    grid = torch.zeros(per_rank * world, dtype=torch.int64, device=dev)
    grid[:n_frames_total] = lengths_all.to(torch.int64)
    grid = grid.view(per_rank, world)
    run_offsets = torch.cumsum(grid, dim=0) - grid            # [j, r]: offset of frame (j, r) inside rank r's run
    if run_capacity is None:
        run_bytes = int(grid.sum(dim=0).max().item())         # host sync: the collective needs one size
    else:
        run_bytes = int(run_capacity)
    cap = (run_bytes + 15) & ~15
    # (exact-size mode zero-fills the run so that the gathered buffer is reproducible byte for byte; in capacity mode
    # the bytes behind a rank's frames are never read -- no memset of hundreds of megabytes per step)
    run = (torch.zeros if run_capacity is None else torch.empty)(max(cap, 16), dtype=torch.uint8, device=dev)
    row_offsets = torch.arange(n_local, dtype=torch.int64, device=dev) * packed.shape[1]
    my_lengths = local_lengths.to(torch.int32).contiguous()
    place(packed, row_offsets, my_lengths, run, run_offsets[:n_local, rank].contiguous())
    if collective is not None:
        runs = torch.empty(world * run.numel(), dtype=torch.uint8, device=dev)
        collective.plain(runs, run)
    elif dist.is_initialized():  # (also a 1-rank group: the same collective call as with 8 ranks)
        runs = torch.empty(world * run.numel(), dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(runs, run, group=group)
    else:
        runs = run
    # frame f = j * world + r lives at r * cap' + run_offsets[j, r] in `runs`
    src = (run_offsets + torch.arange(world, dtype=torch.int64, device=dev) * run.numel()).reshape(-1)[:n_frames_total]
    # capacity mode: the stream cannot be longer than the frames' rows (n_frames_total * row bytes <= world * cap)
    total = int(lengths_all.to(torch.int64).sum().item()) if run_capacity is None else min(world * cap, n_frames_total * packed.shape[1])
    stream_bytes = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
    place(runs, src.contiguous(), lengths_all.to(torch.int32).contiguous(), stream_bytes,
          offsets.to(torch.int64).contiguous())
    return stream_bytes[:total]


def device_place(handle, stream: int | None = None):
    """`place` for all_gather_frame_bytes on the GPU: flacenc_hip_place_frames_async through the C ABI."""
    def place(src, src_offsets, lengths, dst, dst_offsets):
        assert src.is_cuda and dst.is_cuda, "the frame exchange runs on the GPU (no host fallback)"
        handle.place_frames_device(src.data_ptr(), src_offsets.data_ptr(), lengths.data_ptr(), lengths.numel(),
                                   dst.data_ptr(), dst_offsets.data_ptr(), stream=stream)
    return place


def stream_offsets(lengths_all: torch.Tensor, header_bytes: int = 0):
    """Exclusive prefix sum: byte offset of every frame in the output stream (after `header_bytes` of
    container metadata) and the total stream size (a 0-d tensor: no host synchronisation here)."""
    wide = lengths_all.to(torch.int64)
    csum = torch.cumsum(wide, dim=0)
    offsets = csum - wide + header_bytes
    total = csum[-1] + header_bytes if csum.numel() else torch.zeros((), dtype=torch.int64) + header_bytes
    return offsets, total
