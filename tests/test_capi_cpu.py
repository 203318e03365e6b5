"""CPU-only checks of the C-ABI library: it loads, exports every symbol the header declares,
and its host-side entry points behave (no GPU compute is called here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import util
from flacenc_rs_amd import _capi
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _capi.load()
    header = "".join(open(os.path.join(ROOT, "include", f)).read()
                     for f in sorted(os.listdir(os.path.join(ROOT, "include"))) if f.endswith(".h"))
    declared = set(re.findall(r"\b(flacenc_(?:hip|sigen)_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_capi.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.flacenc_hip_abi_version() == _capi.ABI_VERSION == 6


def test_product_library_exports_the_c_abi_and_nothing_else():
    """VERDICT r5 hygiene: libflacenc_hip.so carries no test hook, no C++ internal and no unprefixed helper
    (csrc/exports.map); libflacenc_hip_hooks.so is the same plus the five flacenc_hip_debug_* of csrc/flacenc_hip_debug.h."""
    import subprocess

    def exported(path):
        out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
        return {line.split()[-1] for line in out.splitlines() if line.strip()}

    product, hooks = exported(_capi.LIB_PATH), exported(_capi.HOOKS_LIB_PATH)
    assert product == set(_capi.EXPORTED_SYMBOLS), sorted(product ^ set(_capi.EXPORTED_SYMBOLS))
    assert hooks == product | set(_capi.DEBUG_SYMBOLS), sorted(hooks ^ (product | set(_capi.DEBUG_SYMBOLS)))
    debug_h = open(os.path.join(ROOT, "flacenc_rs_amd", "csrc", "flacenc_hip_debug.h")).read()
    assert set(re.findall(r"\b(flacenc_hip_debug_[a-z0-9_]+)\s*\(", debug_h)) == set(_capi.DEBUG_SYMBOLS)


def test_params_record_layout_matches_oracle_record():
    assert _capi.PARAMS_DTYPE == orc.RECORD_DTYPE
    assert _capi.PARAMS_DTYPE.itemsize == 352


def test_verify_config_mirrors_reference_ranges():
    """config::Qlpc::verify / Prc::verify / Window::verify, src/config.rs:302-326, 224-229, 371-387."""
    ok = _capi.make_config()
    assert _capi.verify_config(ok) == _capi.OK
    for bad in (dict(lpc_order=0), dict(lpc_order=25, flags=0), dict(quant_precision=0),
                dict(quant_precision=16), dict(max_rice_parameter=31), dict(window=("tukey", 1.5)),
                dict(window=("tukey", -0.1))):
        cfg = _capi.make_config(**{k: v for k, v in bad.items() if k != "flags"})
        if "flags" in bad:
            cfg.flags = bad["flags"]
        assert _capi.verify_config(cfg) == _capi.ERR_BAD_CONFIG, bad
    assert _capi.verify_config(_capi.make_config(lpc_order=24)) == _capi.OK
    assert _capi.verify_config(_capi.make_config(lpc_order=32)) == _capi.OK  # extension flag set
    cfg = _capi.make_config()
    cfg.window_type = 7
    assert _capi.verify_config(cfg) == _capi.ERR_BAD_CONFIG
    # ABI 4: the summation-order flags exclude each other, the simd-nightly order stops at lag 15 (above, the split
    # of the reference's buffer depends on its allocator: UNSUPPORTED, not a configuration error), the experimental
    # estimators' fields are accepted (the reference's `experimental` build) within the IRLS step bound
    both = _capi.FLAG_REFERENCE_SUM_ORDER | _capi.FLAG_NIGHTLY_SUM_ORDER
    assert _capi.verify_config(_capi.make_config(flags=both)) == _capi.ERR_BAD_CONFIG
    assert _capi.verify_config(_capi.make_config(lpc_order=15, flags=_capi.FLAG_NIGHTLY_SUM_ORDER)) == _capi.OK
    assert _capi.verify_config(_capi.make_config(lpc_order=16, flags=_capi.FLAG_NIGHTLY_SUM_ORDER)) == _capi.ERR_UNSUPPORTED
    assert _capi.verify_config(_capi.make_config(lpc_order=24, flags=_capi.FLAG_REFERENCE_SUM_ORDER)) == _capi.OK
    assert _capi.verify_config(_capi.make_config(use_direct_mse=True, mae_optimization_steps=3)) == _capi.OK
    assert _capi.verify_config(_capi.make_config(use_direct_mse=True, mae_optimization_steps=65)) == _capi.ERR_BAD_CONFIG


@pytest.mark.parametrize("window,n", [(("tukey", 0.4), 4096), (("tukey", 0.3), 32), (("tukey", 1.0), 1001),
                                      (("tukey", 0.0), 64), ("rectangle", 100), (("tukey", 0.1), 16384)])
def test_window_weights_equal_oracle(window, n):
    """lpc::window_weights (src/lpc.rs:96-120): the product's host table == the oracle's, bitwise."""
    w = _capi.window_weights(_capi.make_config(window=window), n)
    assert np.array_equal(w.view(np.uint32), orc.window_weights(window, n).view(np.uint32))


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_capi.FlacencHipError) as ei:
        _capi.Handle(0)
    assert ei.value.code == _capi.ERR_NO_DEVICE


def test_sigen_matches_numpy_model():
    """flacenc_sigen_fill_frames == Sine + Noise + to_vec_quantized (src/sigen.rs:35-53, 159-168,
    227-232) re-stated in numpy with the same counter-based noise (tests/util.py)."""
    n, bps = 4096, 16
    x = _capi.sigen_frames(3, 2, n, bps, 200.0, 0.4, 0.4, seed=7, first_frame=5)
    assert x.shape == (3, 2, n) and x.dtype == np.int32
    lo, hi = -(1 << (bps - 1)), (1 << (bps - 1)) - 1
    assert x.min() >= lo and x.max() <= hi
    for f in range(3):
        for c in range(2):
            off = (5 + f) * n
            ref = util.quantize(util.sine(n, 200.0 + 7.0 * c, 0.4, phase=0.5 * c, offset=off)
                                + util.noise(7 + c, n, 0.4, offset=off), bps)
            # libm sinf vs numpy's sin may differ in the last ulp -> allow +-1 LSB on a few samples
            d = np.abs(x[f, c] - ref)
            assert d.max() <= 1 and (d != 0).mean() < 0.01
    # determinism and thread-count independence
    y = _capi.sigen_frames(3, 2, n, bps, 200.0, 0.4, 0.4, seed=7, first_frame=5, nthreads=1)
    assert np.array_equal(x, y)


def test_frame_wire_bytes_follow_the_finest_partition_count():
    """flacenc_hip_frame_wire_bytes (host arithmetic): 48 bytes of frame fields + two subframe records cut behind
    2^finest_partition_order Rice parameters (src/rice.rs:157-165, with the oracle's finest_partition_order for the
    largest warm-up the order bound allows) -- the size shard.py's host statement of the wire format uses."""
    from flacenc_rs_amd import shard
    lib = _capi.load()
    for n in list(range(1, 70)) + [96, 100, 128, 192, 256, 288, 512, 576, 1000, 1024, 1152, 2048, 2304, 4095, 4096, 4097,
                                   4608, 8192, 16384, 20000, 32767]:
        assert lib.flacenc_hip_frame_wire_bytes(n) == shard.wire_record_bytes(n), n
        assert shard.wire_record_bytes(n) <= 752
    assert shard.wire_record_bytes(4096) == 368 and shard.wire_record_bytes(16384) == 752
