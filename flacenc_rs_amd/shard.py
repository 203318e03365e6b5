"""Frame sharding across the GPUs of one node and the ordered gather of the results.

The reference's only parallelism is a frame-level worker pool (src/par.rs): frames go to
whichever worker is free and `ParSink` (src/par.rs:67-95) re-orders finished frames by
frame number.  Here the pool is the node's GPUs, one process per GPU: frame f belongs to
rank f mod G (round-robin, BASELINE config 4), every rank analyses its frames with no
communication, and the ordered gather is ONE all-gather of the fixed-size parameter
records (352 B per analysed subframe, include/flacenc_hip.h) over RCCL/xGMI.  Residuals
stay on the GPU that produced them (they are 47x larger than the records and the next
stage, Rice bit-packing, is local to a frame).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def frames_of_rank(n_frames_total: int, rank: int, world: int) -> range:
    """Stream frame numbers owned by `rank`: f = rank, rank + G, rank + 2G, ..."""
    return range(rank, n_frames_total, world)


def local_frame_count(n_frames_total: int, rank: int, world: int) -> int:
    return len(frames_of_rank(n_frames_total, rank, world))


def all_gather_records(local: torch.Tensor, n_frames_total: int, group=None) -> torch.Tensor:
    """All-gather per-frame records and return them in stream (frame-number) order.

    `local` is [n_local_frames, ...] (any trailing shape / dtype) holding this rank's frames in
    the order of `frames_of_rank`.  Ranks may own different frame counts (n_frames_total not a
    multiple of the world size); shorter ranks are padded for the collective.  The result is
    [n_frames_total, ...] on every rank -- what ParSink::finalize hands to the stream writer.
    """
    if not dist.is_available() or not dist.is_initialized():
        assert local.shape[0] == n_frames_total
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per_rank = (n_frames_total + world - 1) // world
    n_local = local_frame_count(n_frames_total, rank, world)
    assert local.shape[0] == n_local, (local.shape, n_local)
    if n_local < per_rank:
        pad = torch.zeros((per_rank - n_local,) + tuple(local.shape[1:]), dtype=local.dtype,
                          device=local.device)
        local = torch.cat([local, pad], dim=0)
    local = local.contiguous()
    gathered = torch.empty((world * per_rank,) + tuple(local.shape[1:]), dtype=local.dtype,
                           device=local.device)
    dist.all_gather_into_tensor(gathered, local, group=group)
    # gathered[r * per_rank + j] is stream frame j * world + r  ->  transpose (r, j) -> (j, r)
    g = gathered.view((world, per_rank) + tuple(local.shape[1:]))
    ordered = g.transpose(0, 1).reshape((world * per_rank,) + tuple(local.shape[1:]))
    return ordered[:n_frames_total]
