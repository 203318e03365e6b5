"""world_size-2 and -8 gloo tests of the multi-GPU path on CPU: round-robin frame sharding + ordered
all-gather of the parameter records must reproduce the single-process result in frame order
(the role of ParSink, src/par.rs:67-95, tested there by `par_sink_finalization`, par.rs:457-556).
No GPU here: each rank fills its records with the CPU oracle (allowed in tests/)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_frames, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from flacenc_rs_amd import _capi, shard
    from oracle import oracle as orc

    n, bps, order = 256, 16, 8
    mine = shard.frames_of_rank(n_frames, rank, world)
    frames = _capi.sigen_frames(len(mine), 2, n, bps, 50.0, 0.4, 0.1, seed=3, first_frame=rank,
                                frame_step=world, nthreads=1)
    recs = np.zeros((len(mine), 2), _capi.PARAMS_DTYPE)
    cfg = orc.make_config(lpc_order=order)
    for j in range(len(mine)):
        r, _, _, _ = orc.qlpc_batch(frames[j], bps, cfg, want_fp=False)
        recs[j] = r
    local = torch.from_numpy(recs.view(np.uint8).reshape(len(mine), 2, 352).copy())
    ordered = shard.all_gather_records(local, n_frames)
    assert ordered.shape == (n_frames, 2, 352)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), ordered.numpy())
    # the light-weight exchange bench.py uses: frame byte lengths -> stream offsets
    lens = torch.tensor([1000 + 7 * f for f in mine], dtype=torch.int32)
    lengths_all = shard.all_gather_frame_lengths(lens, n_frames)
    offsets, total = shard.stream_offsets(lengths_all, header_bytes=42)
    want = np.array([1000 + 7 * f for f in range(n_frames)], np.int64)
    assert lengths_all.tolist() == want.tolist()
    assert offsets.tolist() == (42 + np.concatenate([[0], np.cumsum(want)[:-1]])).tolist()
    assert int(total) == 42 + int(want.sum())
    # the collective alone, as bench.py's exchange step and flacenc_hip_allgather_records_async deliver it and as
    # flacenc_hip_stream_offsets_async (stream_offsets_kernel) reads it: rank-major [world][ceil(F / G)], row
    # r * per_rank + j = stream frame j * G + r, ranks that are a frame short zero-padded
    per_rank = (n_frames + world - 1) // world
    gl = shard.all_gather_rank_major(lens, n_frames)
    assert gl.shape == (world * per_rank,) and gl.dtype == torch.int32
    for r in range(world):
        for j in range(per_rank):
            f = j * world + r
            assert int(gl[r * per_rank + j]) == (1000 + 7 * f if f < n_frames else 0), (r, j)
    # host restatement of stream_offsets_kernel on that layout == stream_offsets on the stream-order lengths
    ls = np.array([int(gl[(f % world) * per_rank + f // world]) for f in range(n_frames)], np.int64)
    assert ls.tolist() == want.tolist()
    assert (42 + np.concatenate([[0], np.cumsum(ls)[:-1]])).tolist() == offsets.tolist()
    # ... and for multi-byte records (the wire records): rows of this rank's frames, zero rows behind a short rank
    rows = torch.stack([torch.full((5,), f % 251, dtype=torch.uint8) for f in mine]) if len(mine) else torch.zeros((0, 5), dtype=torch.uint8)
    gr = shard.all_gather_rank_major(rows, n_frames)
    assert gr.shape == (world * per_rank, 5)
    for r in range(world):
        for j in range(per_rank):
            f = j * world + r
            assert gr[r * per_rank + j].tolist() == ([f % 251] * 5 if f < n_frames else [0] * 5)
    # the heavy exchange (`bench.py --gather payload`): the packed frame bytes themselves, assembled into
    # the frame stream on every rank.  Frames are written by the oracle's Frame::write with their stream
    # frame numbers; `place` here is the test's stand-in for flacenc_hip_place_frames_async.
    fc = orc.make_frame_config(cfg, use_fixed=True)
    res, resid = orc.encode_stereo_frames_cfg(frames, bps, fc)
    blobs = [orc.write_stereo_frame(res[j], frames[j, 0], frames[j, 1], bps, 44100, f, resid[j, 0], resid[j, 1])
             for j, f in enumerate(mine)]
    capt = torch.tensor([max([len(b) for b in blobs] + [0]) + 5], dtype=torch.int64)
    dist.all_reduce(capt, op=dist.ReduceOp.MAX)  # one row width for every rank, like the packer's out_stride
    cap = int(capt.item())
    packed = torch.zeros((len(mine), cap), dtype=torch.uint8)
    for j, b in enumerate(blobs):
        packed[j, :len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8)
    my_len = torch.tensor([len(b) for b in blobs], dtype=torch.int32)

    def place(src, src_offsets, lengths, dst, dst_offsets):
        sflat, dflat = src.reshape(-1), dst.reshape(-1)
        for so, ln, do in zip(src_offsets.tolist(), lengths.tolist(), dst_offsets.tolist()):
            dflat[do:do + ln] = sflat[so:so + ln]

    lengths_all = shard.all_gather_frame_lengths(my_len, n_frames)
    offsets, total = shard.stream_offsets(lengths_all)
    stream = shard.all_gather_frame_bytes(place, packed, my_len, lengths_all, offsets, n_frames)
    assert stream.numel() == int(total)
    np.save(os.path.join(out_dir, f"stream{rank}.npy"), stream.numpy())
    # ... and with `run_capacity` (what bench.py passes: no size is read back; the result is capacity-sized and its
    # first `total` bytes are the stream)
    per_rank = (n_frames + world - 1) // world
    capped = shard.all_gather_frame_bytes(place, packed, my_len, lengths_all, offsets, n_frames, run_capacity=per_rank * cap)
    assert capped.numel() >= int(total) and capped.numel() <= n_frames * cap + 16 * world
    assert torch.equal(capped[:int(total)], stream)
    dist.barrier()
    dist.destroy_process_group()


# (world 8 with a frame count that is not a multiple of 8: the index arithmetic of an 8-GPU node -- ragged last
# round, ranks with one frame fewer -- over gloo; ParSink::finalize, src/par.rs:82-94)
@pytest.mark.parametrize("world,n_frames", [(2, 8), (2, 7), (8, 19), (8, 5)])
def test_round_robin_shard_and_ordered_gather(tmp_path, world, n_frames):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_frames, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from flacenc_rs_amd import _capi
    from oracle import oracle as orc

    # single-process reference: the whole stream in frame order
    frames = _capi.sigen_frames(n_frames, 2, 256, 16, 50.0, 0.4, 0.1, seed=3, nthreads=1)
    want = np.zeros((n_frames, 2), _capi.PARAMS_DTYPE)
    for f in range(n_frames):
        want[f], _, _, _ = orc.qlpc_batch(frames[f], 16, orc.make_config(lpc_order=8), want_fp=False)
    want_bytes = want.view(np.uint8).reshape(n_frames, 2, 352)
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{rank}.npy"))
        assert np.array_equal(got, want_bytes), rank
    # the assembled frame stream: identical on both ranks, and an independent parser walks it frame by
    # frame (numbers 0..n-1 in order, both CRCs) back to the input samples
    from tests import flac_parse
    s0 = np.load(os.path.join(str(tmp_path), "stream0.npy")).tobytes()
    for rank in range(1, world):
        assert s0 == np.load(os.path.join(str(tmp_path), f"stream{rank}.npy")).tobytes()
    pos = 0
    for f in range(n_frames):
        fr = flac_parse.parse_frame(s0[pos:], stream_bps=16, stream_rate=44100)
        assert fr["number"] == f and np.array_equal(fr["channels"], frames[f])
        pos += fr["length"]
    assert pos == len(s0)


def test_frames_of_rank_partition():
    from flacenc_rs_amd import shard
    for total in (0, 1, 7, 8, 9, 64):
        for world in (1, 2, 4, 8):
            seen = sorted(f for r in range(world) for f in shard.frames_of_rank(total, r, world))
            assert seen == list(range(total))
            assert all(f % world == r for r in range(world) for f in shard.frames_of_rank(total, r, world))
