"""Regenerates tests/golden/qlpc_golden.npz with the CPU oracle (reference summation order).

Run from the repo root:  python tests/golden/make_qlpc_golden.py
The vectors are oracle outputs (the Rust reference cannot be built here); they pin the oracle
against drift and give the GPU tests committed expectations.  Inputs are stored alongside the
outputs so that nothing depends on numpy's sin/cos being bit-stable across machines.
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import util  # noqa: E402
from oracle import oracle as orc  # noqa: E402

CASES = [
    # name, n, lpc_order, bps, generator
    ("c2_n4096_p8_b16", 4096, 8, 16, lambda: util.sine_noise(4096, 16, 200, 0.4, 0.4, seed=0xF1AC0001)),
    ("c1_n4096_p10_b16", 4096, 10, 16, lambda: util.sine_noise(4096, 16, 36, 0.4, 0.04, seed=0xF1AC0002)),
    ("c1_n4096_p10_b17_side", 4096, 10, 17,
     lambda: util.sine_noise(4096, 16, 36, 0.4, 0.04, seed=3) - util.sine_noise(4096, 16, 50, 0.3, 0.04, seed=4)),
    ("c3_n8192_p24_b24", 8192, 24, 24, lambda: util.sine_noise(8192, 24, 100, 0.8, 0.2, seed=0xF1AC0003)),
    ("c5_n16384_p24_b24", 16384, 24, 24, lambda: util.sine_noise(16384, 24, 440, 0.5, 0.05, seed=0xF1AC0005)),
    ("noise_n4096_p10_b16", 4096, 10, 16, lambda: util.quantize(util.noise(77, 4096, 0.6), 16)),
    ("sus109_ch0_n4096_p10", 4096, 10, 16, lambda: util.test_signal("sus109", 0)[:4096]),
    ("sus6_ch1_n4096_p8", 4096, 8, 16, lambda: util.test_signal("sus6", 1)[4096:8192]),
    ("ras22_ch0_n1152_p12", 1152, 12, 16, lambda: util.test_signal("ras22", 0)[1000:2152]),
    ("ras103_ch1_n8192_p24", 8192, 24, 16, lambda: util.test_signal("ras103", 1)),
]


def main():
    out = {}
    for name, n, order, bps, gen in CASES:
        x = np.ascontiguousarray(gen(), np.int32)
        assert len(x) == n
        cfg = orc.make_config(lpc_order=order)
        r = orc.estimated_qlpc(x, bps, cfg)
        assert r["status"] == 0
        out[f"{name}/input"] = x.astype(np.int16) if bps <= 16 else x
        out[f"{name}/meta"] = np.array([n, order, bps], np.int64)
        out[f"{name}/autocorr"] = r["autocorr"]
        out[f"{name}/lpc_coefs"] = r["lpc_coefs"]
        out[f"{name}/coefs"] = r["coefs"]
        out[f"{name}/scalars"] = np.array(
            [r["order"], r["shift"], r["rice_order"], r["code_bits"], r["subframe_bits"],
             r["sum_quotients"], zlib.crc32(r["residual"].tobytes())], np.int64)
        out[f"{name}/rice_params"] = r["rice_params"]
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "qlpc_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
