"""End-to-end demo: PCM -> .flac with every per-frame stage on the GPU.

Reads a 16/24-bit stereo WAV (or synthesises one), cuts it into 4096-sample frames, runs
flacenc_hip_encode_stereo_frames (analysis + encode_frame's decisions) and
flacenc_hip_pack_stereo_frames (Frame::write), and writes "fLaC" + STREAMINFO + the frames.
What stays on the host is what the reference keeps serial too: the container's 42 bytes and the
MD5 of the input (src/source.rs:406-428).  A short tail block is a second (one-frame) batch: the
frame-level entry points take any block size.

    python tools/encode_flac.py [in.wav] out.flac [--seconds 10]
"""
import argparse
import hashlib
import os
import struct
import sys
import wave

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from flacenc_rs_amd import _capi  # noqa: E402


def stream_info_block(block_size, min_frame, max_frame, rate, channels, bps, total, md5):
    """MetadataBlock(StreamInfo)::write, src/component/bitrepr.rs:199-270 (last-block flag set)."""
    body = struct.pack(">HH", block_size, block_size)
    body += min_frame.to_bytes(3, "big") + max_frame.to_bytes(3, "big")
    packed = (rate << 44) | ((channels - 1) << 41) | ((bps - 1) << 36) | total
    body += packed.to_bytes(8, "big") + md5
    return bytes([0x80]) + len(body).to_bytes(3, "big") + body


def md5_of(frames, bps):
    """Source MD5: interleaved little-endian samples of ceil(bps / 8) bytes, src/source.rs:406-428."""
    inter = np.ascontiguousarray(frames.transpose(0, 2, 1)).reshape(-1)
    nbytes = (bps + 7) // 8
    raw = inter.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :nbytes]
    return hashlib.md5(raw.tobytes()).digest()


def encode_pcm(pcm, bps, rate, handle, block_size=4096, use_fixed=True, lpc_order=8):
    """pcm int32 [n_samples, 2] -> (.flac bytes, per-frame decision records).  Like
    encode_with_fixed_block_size (src/coding.rs:645-700): whole blocks, then the shorter last one."""
    cfg = _capi.make_frame_config(_capi.make_config(lpc_order=lpc_order), use_fixed=use_fixed)
    n_full = pcm.shape[0] // block_size
    groups = []
    if n_full:
        groups.append(np.ascontiguousarray(pcm[: n_full * block_size].reshape(n_full, block_size, 2).transpose(0, 2, 1)))
    tail = pcm.shape[0] - n_full * block_size
    if tail:
        groups.append(np.ascontiguousarray(pcm[n_full * block_size:].T[None]))
    packed, records, first = [], [], 0
    for g in groups:
        if g.shape[2] < 64:   # blocks below MIN_BLOCK_SIZE_FOR_PREDICTION (constant.rs:51) never reach the GPU path
            raise ValueError("a tail block shorter than 64 samples is not supported by this demo")
        res, resid = handle.encode_stereo_frames(g, bps, cfg)
        packed += handle.pack_stereo_frames(g, res, resid, bps, rate, first_frame_number=first)
        records.append(res)
        first += g.shape[0]
    inter = np.ascontiguousarray(pcm).reshape(-1)
    nbytes = (bps + 7) // 8
    md5 = hashlib.md5(inter.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :nbytes].tobytes()).digest()
    sizes = list(map(len, packed))
    # fixed-block streams announce the block size as both min and max (src/coding.rs:676-690)
    head = b"fLaC" + stream_info_block(block_size, min(sizes), max(sizes), rate, 2, bps, pcm.shape[0], md5)
    return head + b"".join(packed), np.concatenate(records)


def encode(frames, bps, rate, handle, use_fixed=True, lpc_order=8):
    """frames int32 [n_frames, 2, n] (whole blocks only) -> (.flac bytes, records)."""
    pcm = np.ascontiguousarray(frames.transpose(0, 2, 1)).reshape(-1, 2)
    return encode_pcm(pcm, bps, rate, handle, block_size=frames.shape[2], use_fixed=use_fixed, lpc_order=lpc_order)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("paths", nargs="+")
    ap.add_argument("--seconds", type=float, default=10.0)
    args = ap.parse_args()
    n = 4096
    if len(args.paths) == 2:
        with wave.open(args.paths[0], "rb") as w:
            assert w.getnchannels() == 2 and w.getsampwidth() in (2, 3)
            bps, rate = 8 * w.getsampwidth(), w.getframerate()
            raw = np.frombuffer(w.readframes(w.getnframes()), np.uint8).reshape(-1, 2, w.getsampwidth())
        pad = np.zeros(raw.shape[:2] + (4 - raw.shape[2],), np.uint8)
        pcm = (np.concatenate([pad, raw], axis=2).view("<i4")[..., 0] >> (32 - bps)).astype(np.int32)
    else:
        bps, rate = 16, 44100
        nsamp = max(n, int(args.seconds * rate))
        nf = (nsamp + n - 1) // n
        pcm = np.ascontiguousarray(_capi.sigen_frames(nf, 2, n, bps, rate / 440.0, 0.8, 0.2, seed=1)
                                   .transpose(0, 2, 1)).reshape(-1, 2)[:nsamp]
    data, res = encode_pcm(pcm, bps, rate, _capi.Handle(0), block_size=n)
    nf = len(res)
    with open(args.paths[-1], "wb") as f:
        f.write(data)
    kinds = np.bincount(res["kind"].ravel(), minlength=4)
    print(f"{nf} frames, {len(data)} bytes, {len(data) / (pcm.shape[0] * 2 * bps / 8):.4f} of the PCM size; "
          f"subframes constant/verbatim/fixed/lpc = {kinds.tolist()}")


if __name__ == "__main__":
    main()
