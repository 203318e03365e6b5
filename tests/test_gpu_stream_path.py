"""flacenc_hip_encode_pcm_stereo: packed interleaved PCM in host memory -> FLAC frame bytes in host memory,
chunked and double-buffered over two copy streams.  The bytes must be exactly what the one-call device path
(and therefore the oracle's controller + bit writer, see test_gpu_parity.py) produces for the same frames,
whatever the chunking, for pageable and for page-locked caller buffers, with a short last block, with 16-
and 24-bit samples, and with frame numbers dealt round-robin (first / step)."""
import numpy as np
import pytest

import flac_parse
from flacenc_rs_amd import _capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def handle():
    h = _capi.Handle(0)
    yield h
    h.close()


def pack_pcm(frames_i32, bytes_per_sample):
    """int32 [F, 2, n] (+ optional tail [2, m]) -> interleaved little-endian bytes"""
    inter = np.ascontiguousarray(frames_i32.transpose(0, 2, 1)).reshape(-1, 2)  # [F*n, 2]
    raw = inter.astype("<i4").view(np.uint8).reshape(-1, 2, 4)[:, :, :bytes_per_sample]
    return np.ascontiguousarray(raw).reshape(-1)


def reference_bytes(handle, frames, bps, cfg, rate, first=0, step=1):
    res, resid = handle.encode_stereo_frames(frames, bps, cfg)
    return handle.pack_stereo_frames(frames, res, resid, bps, rate, first, step)


@pytest.mark.parametrize("bps,bytes_ps,n,F,use_fixed", [(16, 2, 4096, 9, True), (16, 2, 4096, 1700, False),
                                                        (24, 3, 4096, 5, True), (16, 2, 1152, 40, True),
                                                        (24, 3, 8192, 6, False)])
def test_stream_bytes_equal_the_one_call_path(handle, bps, bytes_ps, n, F, use_fixed):
    frames = _capi.sigen_frames(F, 2, n, bps, 50.0, 0.3, 0.05, seed=1234 + n + F, nthreads=4)
    cfg = _capi.make_frame_config(_capi.make_config(lpc_order=8 if n == 4096 else 12), use_fixed=use_fixed)
    pcm = pack_pcm(frames, bytes_ps)
    out, lens = handle.encode_pcm_stereo(pcm, cfg, bytes_ps, bps, n, 44100)
    want = reference_bytes(handle, frames[: min(F, 64)], bps, cfg, 44100)
    assert lens[: len(want)].tolist() == [len(b) for b in want]
    pos = 0
    for f, b in enumerate(want):
        assert out[pos:pos + len(b)].tobytes() == b, f
        pos += len(b)
    assert int(lens.astype(np.int64).sum()) == out.size
    # every frame of the stream parses (sync, both CRCs) with its number, and decodes to the input
    pos = 0
    for f in list(range(min(F, 3))) + ([F - 1] if F > 3 else []):
        start = int(lens[:f].astype(np.int64).sum())
        fr = flac_parse.parse_frame(out[start:start + int(lens[f])].tobytes(), stream_bps=bps, stream_rate=44100)
        assert fr["number"] == f and np.array_equal(fr["channels"], frames[f])


def test_short_last_block_and_round_robin_numbers(handle):
    n, bps = 4096, 16
    frames = _capi.sigen_frames(5, 2, n, bps, 36.0, 0.4, 0.04, seed=77, nthreads=2)
    tail = frames[4][:, :1000]
    cfg = _capi.make_frame_config(_capi.make_config(lpc_order=10), use_fixed=True)
    pcm = np.concatenate([pack_pcm(frames[:4], 2), pack_pcm(tail[None], 2)])
    out, lens = handle.encode_pcm_stereo(pcm, cfg, 2, bps, n, 48000, first_frame_number=3, frame_number_step=8)
    assert lens.size == 5
    want = reference_bytes(handle, frames[:4], bps, cfg, 48000, 3, 8) + reference_bytes(handle, np.ascontiguousarray(tail[None]), bps, cfg, 48000, 3 + 4 * 8, 8)
    assert out.tobytes() == b"".join(want)
    last = flac_parse.parse_frame(want[-1], stream_bps=bps, stream_rate=48000)
    assert last["number"] == 35 and last["block_size"] == 1000 and np.array_equal(last["channels"], tail)


@pytest.mark.parametrize("tail_len", [1, 10, 63])
def test_last_block_shorter_than_64_samples(handle, tail_len):
    """A last block below MIN_BLOCK_SIZE_FOR_PREDICTION is a frame like any other: encode_subframe skips its
    predictors (too_short, src/coding.rs:389-418) and emits Constant or Verbatim, the stereo decision still runs."""
    n, bps = 4096, 16
    frames = _capi.sigen_frames(3, 2, n, bps, 36.0, 0.4, 0.04, seed=78, nthreads=2)
    tail = np.ascontiguousarray(frames[2][:, :tail_len])
    if tail_len == 10:
        tail[1] = 7  # a Constant subframe
    cfg = _capi.make_frame_config(_capi.make_config(lpc_order=10), use_fixed=True)
    pcm = np.concatenate([pack_pcm(frames[:2], 2), pack_pcm(tail[None], 2)])
    out, lens = handle.encode_pcm_stereo(pcm, cfg, 2, bps, n, 44100)
    assert lens.size == 3
    want = reference_bytes(handle, frames[:2], bps, cfg, 44100) + reference_bytes(handle, tail[None], bps, cfg, 44100, 2, 1)
    assert out.tobytes() == b"".join(want)
    last = flac_parse.parse_frame(want[-1], stream_bps=bps, stream_rate=44100)
    assert last["number"] == 2 and last["block_size"] == tail_len and np.array_equal(last["channels"], tail)


def test_pinned_buffers_give_identical_bytes(handle):
    n, bps, F = 4096, 16, 1100   # more than one chunk
    frames = _capi.sigen_frames(F, 2, n, bps, 200.0, 0.4, 0.4, seed=5, nthreads=4)
    cfg = _capi.make_frame_config(_capi.make_config(lpc_order=8), use_fixed=False)
    pcm = pack_pcm(frames, 2)
    out_a, lens_a = handle.encode_pcm_stereo(pcm, cfg, 2, bps, n, 44100)
    pin_in = _capi.pinned_array(pcm.size)
    pin_in[:] = pcm
    pin_out = _capi.pinned_array(F * (handle.frame_bytes_bound(n, bps) + 16))
    out_b, lens_b = handle.encode_pcm_stereo(pin_in, cfg, 2, bps, n, 44100, out=pin_out)
    assert np.array_equal(lens_a, lens_b) and out_a.tobytes() == out_b.tobytes()
    # twice through the same handle (staging buffers reused)
    out_c, lens_c = handle.encode_pcm_stereo(pcm, cfg, 2, bps, n, 44100)
    assert out_c.tobytes() == out_a.tobytes()


@pytest.mark.parametrize("threads", [0, 1, 3, 8])
def test_staging_thread_counts_give_identical_bytes(handle, threads):
    """flacenc_hip_set_host_threads: the staging copies of pageable buffers are cut into slices for helper
    threads; three chunks in flight (the two-step way out: a chunk's transfer runs during the next chunk's
    staging copy in), a last chunk that is not full."""
    n, bps, F = 4096, 16, 8192 * 2 + 300
    frames = _capi.sigen_frames(F, 2, n, bps, 90.0, 0.3, 0.2, seed=99, nthreads=4)
    cfg = _capi.make_frame_config(_capi.make_config(lpc_order=8), use_fixed=False)
    pcm = pack_pcm(frames, 2)
    handle.set_host_threads(4)
    want, want_lens = handle.encode_pcm_stereo(pcm, cfg, 2, bps, n, 44100)
    try:
        handle.set_host_threads(threads)
        out, lens = handle.encode_pcm_stereo(pcm, cfg, 2, bps, n, 44100)
    finally:
        handle.set_host_threads(4)
    assert np.array_equal(lens, want_lens) and out.tobytes() == want.tobytes()
    first = reference_bytes(handle, frames[:8], bps, cfg, 44100)
    assert out[: sum(len(b) for b in first)].tobytes() == b"".join(first)
    last = reference_bytes(handle, frames[F - 4:], bps, cfg, 44100, F - 4, 1)
    assert out[out.size - sum(len(b) for b in last):].tobytes() == b"".join(last)
    with pytest.raises(_capi.FlacencHipError):
        handle.set_host_threads(-1)


@pytest.mark.parametrize("n,order", [(4096, 8), (4608, 10), (8192, 24), (1152, 8)])
def test_record_wire_format_is_lossless(handle, n, order):
    """The multi-GPU exchange moves frame records without the always-zero tails of their Rice-parameter
    arrays (flacenc_rs_amd/shard.py): real records of every partition-count class must survive the round trip."""
    import torch
    from flacenc_rs_amd import shard
    frames = _capi.sigen_frames(7, 2, n, 16, 36.0, 0.4, 0.2, seed=n + order, nthreads=2)
    res, _ = handle.encode_stereo_frames(frames, 16, _capi.make_frame_config(_capi.make_config(lpc_order=order), use_fixed=True))
    rec = torch.from_numpy(np.frombuffer(res.tobytes(), np.uint8).reshape(7, 752).copy())
    wire = shard.records_to_wire(rec, n)
    assert wire.shape == (7, shard.wire_record_bytes(n)) and wire.shape[1] < 752
    assert torch.equal(shard.records_from_wire(wire, n), rec)
    assert int(res["lpc"]["rice_order"].max()) <= int(np.log2(shard.finest_partitions(n)))


@pytest.mark.parametrize("channels,F", [(1, 5), (3, 4), (8, 1100)])
def test_stream_path_for_independent_channel_frames(handle, channels, F):
    """flacenc_hip_encode_pcm for mono / multi-channel streams (BASELINE configs[3] is the 8-channel case):
    bytes equal to encode_frames + pack_frames of the same frames, short tail block included."""
    n, bps = 4096, 16
    frames = _capi.sigen_frames(F, channels, n, bps, 50.0, 0.3, 0.05, seed=900 + channels, nthreads=4)
    tail = frames[-1][:, :777]
    cfg = _capi.make_frame_config(_capi.make_config(lpc_order=10), use_fixed=True)
    inter = np.concatenate([np.ascontiguousarray(frames[:-1].transpose(0, 2, 1)).reshape(-1, channels), tail.T])
    pcm = np.ascontiguousarray(inter.astype("<i4").view(np.uint8).reshape(-1, channels, 4)[:, :, :2]).reshape(-1)
    out, lens = handle.encode_pcm(pcm, channels, cfg, 2, bps, n, 48000)
    assert lens.size == F
    k = min(F - 1, 40)
    res, resid = handle.encode_frames(frames[:k], bps, cfg)
    want = handle.pack_frames(frames[:k], res, resid, bps, 48000)
    assert lens[:k].tolist() == [len(b) for b in want]
    assert out[: int(lens[:k].astype(np.int64).sum())].tobytes() == b"".join(want)
    tres, tresid = handle.encode_frames(np.ascontiguousarray(tail[None]), bps, cfg)
    twant = handle.pack_frames(np.ascontiguousarray(tail[None]), tres, tresid, bps, 48000, first_frame_number=F - 1)
    assert out[out.size - int(lens[-1]):].tobytes() == twant[0]
    fr = flac_parse.parse_frame(twant[0], stream_bps=bps, stream_rate=48000)
    assert fr["number"] == F - 1 and fr["block_size"] == 777 and np.array_equal(fr["channels"], tail)


def test_small_output_buffer_is_an_error_and_the_handle_survives(handle):
    n, bps, F = 4096, 16, 900
    frames = _capi.sigen_frames(F, 2, n, bps, 200.0, 0.4, 0.4, seed=11, nthreads=4)
    cfg = _capi.make_frame_config(_capi.make_config(lpc_order=8), use_fixed=False)
    pcm = pack_pcm(frames, 2)
    with pytest.raises(_capi.FlacencHipError) as ei:
        handle.encode_pcm_stereo(pcm, cfg, 2, bps, n, 44100, out=np.empty(100000, np.uint8))
    assert ei.value.code == _capi.ERR_BAD_ARGUMENT
    out, lens = handle.encode_pcm_stereo(pcm, cfg, 2, bps, n, 44100)   # same handle, right away
    assert lens.size == F and int(lens.astype(np.int64).sum()) == out.size
