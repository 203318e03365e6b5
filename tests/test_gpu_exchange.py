"""The ordered gather's two device kernels (ParSink, src/par.rs:67-95, across GPUs) against the host-side statements of
the same steps in flacenc_rs_amd/shard.py: records -> wire records + byte lengths (flacenc_hip_stereo_frame_wire_async)
and all-gathered lengths -> stream offsets (flacenc_hip_stream_offsets_async).  Byte work: bit-exact."""
import numpy as np
import pytest

from flacenc_rs_amd import _capi, shard

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def handle():
    return _capi.Handle(0)


def _records(handle, torch, n_frames, block, bps, seed):
    frames = _capi.sigen_frames(n_frames, 2, block, bps, 120.0, 0.4, 0.3, seed=seed)
    x = torch.from_numpy(frames).cuda()
    cfg = _capi.make_frame_config(_capi.make_config(lpc_order=8), use_fixed=True)
    results = torch.empty((n_frames, _capi.FRAME_RESULT_DTYPE.itemsize), dtype=torch.uint8, device="cuda")
    residual = torch.empty((n_frames * 2, block), dtype=torch.int32, device="cuda")
    handle.encode_stereo_frames_device(cfg, x.data_ptr(), n_frames, block, block, bps, results.data_ptr(),
                                       residual.data_ptr(), block)
    torch.cuda.synchronize()
    return results


# blocks whose finest partition count is a multiple of 16 (16-byte units), of 4 (dwords) and neither (bytes)
@pytest.mark.parametrize("block,bps,n_frames", [(4096, 16, 300), (1152, 16, 65), (256, 16, 33), (192, 24, 17),
                                                (16384, 24, 9), (64, 8, 5)])
def test_wire_records_and_lengths_match_the_host_statement(handle, block, bps, n_frames):
    import torch
    results = _records(handle, torch, n_frames, block, bps, seed=block + n_frames)
    first, step = 3, 8  # rank 3 of 8
    wire, lengths = shard.records_to_wire_device(handle, results, block, bps, 44100, first, step)
    want_len = torch.empty(n_frames, dtype=torch.int32, device="cuda")
    handle.stereo_frame_lengths_device(results.data_ptr(), n_frames, block, bps, 44100, first, step, want_len.data_ptr())
    torch.cuda.synchronize()
    assert wire.shape == (n_frames, shard.wire_record_bytes(block))
    assert torch.equal(wire, shard.records_to_wire(results, block))
    assert torch.equal(lengths, want_len)
    # and back: the wire form loses nothing (every dropped byte of rice_params is zero)
    assert torch.equal(shard.records_from_wire(wire, block), results)
    # lengths are optional
    wire2 = torch.zeros_like(wire)
    handle.stereo_frame_wire_device(results.data_ptr(), n_frames, block, bps, 44100, first, step, wire2.data_ptr(),
                                    wire2.shape[1], None)
    torch.cuda.synchronize()
    assert torch.equal(wire2, wire)


def test_wire_rejects_a_short_stride(handle):
    import torch
    results = torch.zeros((4, 752), dtype=torch.uint8, device="cuda")
    wire = torch.zeros((4, 368), dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError):
        handle.stereo_frame_wire_device(results.data_ptr(), 4, 4096, 16, 44100, 0, 1, wire.data_ptr(), 352, None)


@pytest.mark.parametrize("world,n_total", [(1, 1), (1, 4096), (1, 24576), (8, 19), (8, 5), (2, 8193), (8, 196608),
                                           (3, 12289), (8, 4097)])
def test_stream_offsets_from_the_gathered_layout(handle, world, n_total):
    import torch
    rng = np.random.default_rng(world * 1000003 + n_total)
    lengths = rng.integers(14, 1 << 16, size=n_total, dtype=np.int64).astype(np.int32)
    if n_total > 100:
        lengths[7] = 0x7FFFFFF0  # sums pass 2^32: offsets are 64-bit
        lengths[n_total - 2] = 0x7FFFFFF0
    per_rank = (n_total + world - 1) // world
    gathered = np.zeros((world, per_rank), np.int32)  # what all_gather_into_tensor delivers: rank-major, zero-padded
    for r in range(world):
        mine = lengths[r::world]
        gathered[r, :len(mine)] = mine
    g = torch.from_numpy(gathered.reshape(-1)).cuda()
    lengths_all, offsets, total = shard.stream_offsets_device(handle, g, n_total, world, header_bytes=42)
    torch.cuda.synchronize()
    want_off, want_total = shard.stream_offsets(torch.from_numpy(lengths), header_bytes=42)
    assert lengths_all.cpu().numpy().tolist() == lengths.tolist()
    assert torch.equal(offsets.cpu(), want_off)
    assert int(total) == int(want_total)


def test_stream_offsets_of_an_empty_stream(handle):
    import torch
    total = torch.full((), -1, dtype=torch.int64, device="cuda")
    handle.stream_offsets_device(None, 0, 4, 42, None, None, total.data_ptr())
    torch.cuda.synchronize()
    assert int(total) == 42


# ---- the collective inside the C library (flacenc_hip_comm_* / flacenc_hip_allgather_*): ParSink's ordered gather
# (src/par.rs:67-95) for a Rust / C++ host with one process per GPU.  A test box has one GPU: a 1-rank RCCL
# communicator makes exactly the calls an 8-rank one makes.
def test_one_rank_communicator_gathers_like_shard(handle):
    import torch
    n_frames, block, bps = 77, 4096, 16
    results = _records(handle, torch, n_frames, block, bps, seed=4242)
    wire, lengths = shard.records_to_wire_device(handle, results, block, bps, 44100, 0, 1)
    h2 = _capi.Handle(0)
    assert h2.comm_info() == (0, 0)
    uid = _capi.Handle.comm_unique_id()
    assert len(uid) == _capi.COMM_ID_BYTES and any(uid)
    h2.comm_create(uid, 0, 1)
    assert h2.comm_info() == (0, 1)
    with pytest.raises(RuntimeError):
        h2.comm_create(uid, 0, 1)  # one communicator per handle
    stream = torch.cuda.current_stream().cuda_stream
    wb = wire.shape[1]
    gathered = torch.full((n_frames, wb), 0xAB, dtype=torch.uint8, device="cuda")
    h2.allgather_records_device(wire.data_ptr(), n_frames, n_frames, wb, gathered.data_ptr(), stream=stream)
    glen = torch.full((n_frames,), -1, dtype=torch.int32, device="cuda")
    h2.allgather_records_device(lengths.data_ptr(), n_frames, n_frames, 4, glen.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    # what shard.py's statement of the same exchange delivers (no process group: world 1)
    assert torch.equal(gathered, shard.all_gather_rank_major(wire, n_frames))
    assert torch.equal(gathered, shard.all_gather_records(wire, n_frames))
    assert torch.equal(glen, shard.all_gather_rank_major(lengths, n_frames))
    # in place (the rank's own slot is the send buffer) and the plain byte all-gather
    inplace = wire.clone()
    h2.allgather_records_device(inplace.data_ptr(), n_frames, n_frames, wb, inplace.data_ptr(), stream=stream)
    plain = torch.zeros_like(wire)
    h2.allgather_device(wire.data_ptr(), plain.data_ptr(), wire.numel(), stream=stream)
    torch.cuda.synchronize()
    assert torch.equal(inplace, wire) and torch.equal(plain, wire)
    # the gathered lengths feed flacenc_hip_stream_offsets_async as they are
    lengths_all, offsets, total = shard.stream_offsets_device(h2, glen, n_frames, 1)
    torch.cuda.synchronize()
    want = torch.cumsum(lengths.to(torch.int64), 0) - lengths.to(torch.int64)
    assert torch.equal(offsets, want) and int(total.item()) == int(lengths.to(torch.int64).sum().item())
    # a count that is not this rank's share of the total is refused
    with pytest.raises(RuntimeError):
        h2.allgather_records_device(wire.data_ptr(), n_frames - 1, n_frames, wb, gathered.data_ptr(), stream=stream)
    h2.comm_destroy()
    assert h2.comm_info() == (0, 0)
    with pytest.raises(RuntimeError):
        h2.allgather_device(wire.data_ptr(), plain.data_ptr(), 16, stream=stream)


def test_allgather_without_a_communicator_is_an_error(handle):
    import torch
    x = torch.zeros(64, dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError):
        handle.allgather_records_device(x.data_ptr(), 4, 4, 16, x.data_ptr())
