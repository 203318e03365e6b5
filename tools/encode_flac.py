"""End-to-end demo: PCM -> .flac with every per-frame stage on the GPU.

Reads a 16/24-bit stereo WAV (or synthesises one), cuts it into 4096-sample frames, runs
flacenc_hip_encode_stereo_frames (analysis + encode_frame's decisions) and
flacenc_hip_pack_stereo_frames (Frame::write), and writes "fLaC" + STREAMINFO + the frames.
What stays on the host is what the reference keeps serial too: the container's 42 bytes and the
MD5 of the input (src/source.rs:406-428).  The input is truncated to whole frames: the short tail
block of a stream goes through the candidate-level entry points.

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


def encode(frames, bps, rate, handle, use_fixed=True, lpc_order=8):
    cfg = _capi.make_frame_config(_capi.make_config(lpc_order=lpc_order), use_fixed=use_fixed)
    res, resid = handle.encode_stereo_frames(frames, bps, cfg)
    packed = handle.pack_stereo_frames(frames, res, resid, bps, rate)
    n = frames.shape[2]
    head = b"fLaC" + stream_info_block(n, min(map(len, packed)), max(map(len, packed)), rate, 2, bps,
                                       frames.shape[0] * n, md5_of(frames, bps))
    return head + b"".join(packed), res


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
        nf = pcm.shape[0] // n
        frames = np.ascontiguousarray(pcm[: nf * n].reshape(nf, n, 2).transpose(0, 2, 1))
    else:
        bps, rate = 16, 44100
        nf = max(1, int(args.seconds * rate) // n)
        frames = _capi.sigen_frames(nf, 2, n, bps, rate / 440.0, 0.8, 0.2, seed=1)
    data, res = encode(frames, bps, rate, _capi.Handle(0))
    with open(args.paths[-1], "wb") as f:
        f.write(data)
    kinds = np.bincount(res["kind"].ravel(), minlength=4)
    print(f"{nf} frames, {len(data)} bytes, {len(data) / (nf * n * 2 * bps / 8):.4f} of the PCM size; "
          f"subframes constant/verbatim/fixed/lpc = {kinds.tolist()}")


if __name__ == "__main__":
    main()
