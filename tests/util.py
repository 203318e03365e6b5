"""Shared test helpers: deterministic sigen-like signals (numpy only) and fixture loading."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """Vectorised splitmix64 finaliser (counter-based RNG; identical in csrc/sigen)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def noise(seed, n, amplitude, offset=0):
    """Uniform noise in (-amp, amp), like sigen::Noise (src/sigen.rs:227-232) but with a
    counter-based generator (rand's StdRng stream is not reproducible here)."""
    with np.errstate(over="ignore"):
        ctr = np.uint64(seed) * np.uint64(0x100000001B3) + np.arange(offset, offset + n, dtype=np.uint64)
    bits = splitmix64(ctr) >> np.uint64(40)  # 24 random bits
    u = (bits.astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / (1 << 24))  # Open01
    return (np.float32(amplitude) * np.float32(2.0) * (u - np.float32(0.5))).astype(np.float32)


def sine(n, period, amplitude, phase=0.0, offset=0):
    """sigen::Sine (src/sigen.rs:159-168), f32 arithmetic."""
    t = np.arange(offset, offset + n).astype(np.float32)
    arg = np.float32(phase) + np.float32(2.0) * np.float32(np.pi) * t / np.float32(period)
    return (np.float32(amplitude) * np.sin(arg.astype(np.float32))).astype(np.float32)


def quantize(x, bits_per_sample):
    """Signal::to_vec_quantized (src/sigen.rs:35-53): scale, round half away, clamp."""
    scale = np.float32(1 << (bits_per_sample - 1))
    v = (scale * np.asarray(x, np.float32)).astype(np.float32)
    r = np.where(v >= 0, np.floor(v + np.float32(0.5)), np.ceil(v - np.float32(0.5)))
    lo, hi = -(1 << (bits_per_sample - 1)), (1 << (bits_per_sample - 1)) - 1
    return np.clip(r, lo, hi).astype(np.int32)


def sine_noise(n, bps, period, amp, namp, seed, phase=0.0):
    return quantize(sine(n, period, amp, phase) + noise(seed, n, namp), bps)


def test_signal(name, ch):
    """test_helper::test_signal (src/test_helper.rs:81-125): 8192 x i16 LE -> i32."""
    path = os.path.join(GOLDEN, f"testsignal.{name}.ch{ch}.bin")
    return np.fromfile(path, dtype="<i2").astype(np.int32)


def residual_write_bits(res):
    """Bit length `Residual::write` produces (src/component/bitrepr.rs:550-597), by simulation:
    6 header bits, 4|5 bits per partition parameter, q zeros + (p+1) bits per coded sample."""
    order = res["partition_order"]
    nparts = 1 << order
    ps = np.asarray(res["rice_params"][:nparts], np.int64)
    param_bits = 5 if (ps > 14).any() else 4
    n, warm = res["block_size"], res["warmup_length"]
    part_len = n >> order
    total = 6 + nparts * param_bits
    q = np.asarray(res["quotients"], np.int64)
    for p in range(nparts):
        start = max(warm, p * part_len)
        end = (p + 1) * part_len
        total += int(q[start:end].sum()) + (end - start) * (int(ps[p]) + 1)
    return total
