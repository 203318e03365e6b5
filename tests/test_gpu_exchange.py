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
