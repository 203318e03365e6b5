"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module; the product package never does.  See
``oracle/flacenc_oracle.h`` for what the oracle is and how it is pinned.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libflacenc_oracle.so")

ACORR_REFERENCE = 0
ACORR_CANONICAL = 1
ACORR_NIGHTLY = 2
ACORR_CHUNK_TREE = 4  # the chunk tree without the certificate (FLACENC_HIP_FLAG_CANONICAL_SUM_ORDER)
ACORR_GENERIC_TREE = 5  # the 16-sample chunk tree on every shape
ACORR_DIRECT_MSE = 3  # config::Qlpc::use_direct_mse; mae_optimization_steps in bits 8.. (experimental, X1)
SUMABS_STABLE = 0
SUMABS_NIGHTLY = 1
SUMABS_CANONICAL = 2
ORDERSEL_BITCOUNT = 0
ORDERSEL_APPROXENT = 1
KIND_CONSTANT, KIND_VERBATIM, KIND_FIXED, KIND_LPC = 0, 1, 2, 3
FIXED_LPC_COEFS = [[0, 0, 0, 0], [1, 0, 0, 0], [2, -1, 0, 0], [3, -3, 1, 0], [4, -6, 4, -1]]
WINDOW_RECTANGLE = 0
WINDOW_TUKEY = 1
MAX_P_TO_BITS = (1 << 27) - 1


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (``make -C oracle``)."""
    src = os.path.join(_HERE, "flacenc_oracle.c")
    hdr = os.path.join(_HERE, "flacenc_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, hdr)
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libflacenc_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


class QlpcConfig(C.Structure):
    _fields_ = [
        ("lpc_order", C.c_uint32),
        ("quant_precision", C.c_uint32),
        ("window_type", C.c_uint32),
        ("tukey_alpha", C.c_float),
        ("max_rice_parameter", C.c_uint32),
        ("acorr_order", C.c_uint32),
    ]


class FixedConfig(C.Structure):
    """config::Fixed, src/config.rs:236-244."""
    _fields_ = [("max_order", C.c_uint32), ("order_sel", C.c_uint32), ("partitions", C.c_uint32),
                ("sum_mode", C.c_uint32)]


class FrameConfig(C.Structure):
    """config::SubFrameCoding + config::StereoCoding, src/config.rs:167-183, 137-144."""
    _fields_ = [("qlpc", QlpcConfig), ("use_constant", C.c_uint32), ("use_fixed", C.c_uint32),
                ("use_lpc", C.c_uint32), ("use_leftside", C.c_uint32), ("use_rightside", C.c_uint32),
                ("use_midside", C.c_uint32), ("fixed", FixedConfig)]


class FixedResult(C.Structure):
    _fields_ = [("selected", C.c_int32), ("order", C.c_uint32), ("estimate", C.c_uint64 * 5),
                ("rice_order", C.c_uint32), ("code_bits", C.c_uint64), ("sum_quotients", C.c_uint64),
                ("sum_rice_params", C.c_uint64), ("residual_bits", C.c_uint64),
                ("subframe_bits", C.c_uint64)]


class SubframeDesc(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("bps", C.c_uint32), ("dc_offset", C.c_int32),
                ("samples", C.POINTER(C.c_int32)), ("order", C.c_uint32), ("shift", C.c_int32),
                ("precision", C.c_uint32), ("coefs", C.POINTER(C.c_int16)), ("rice_order", C.c_uint32),
                ("rice_params", C.POINTER(C.c_uint8)), ("residual", C.POINTER(C.c_int32))]


class QParams(C.Structure):
    _fields_ = [
        ("coefs", C.c_int16 * 32),
        ("order", C.c_uint32),
        ("shift", C.c_int32),
        ("precision", C.c_uint32),
    ]


class PrcParameter(C.Structure):
    _fields_ = [
        ("order", C.c_uint32),
        ("code_bits", C.c_uint64),
        ("ps", C.c_uint8 * (256 * 128)),
    ]


class QlpcResult(C.Structure):
    _fields_ = [
        ("qp", QParams),
        ("rice_order", C.c_uint32),
        ("code_bits", C.c_uint64),
        ("sum_quotients", C.c_uint64),
        ("sum_rice_params", C.c_uint64),
        ("residual_bits", C.c_uint64),
        ("subframe_bits", C.c_uint64),
        ("status", C.c_int32),
        ("autocorr", C.c_double * 33),
        ("lpc_coefs", C.c_double * 32),
    ]


# numpy view of orc_subframe_record == flacenc_hip_subframe_params (352 bytes)
RECORD_DTYPE = np.dtype(
    [
        ("coefs", np.int16, (32,)),
        ("order", np.uint8),
        ("shift", np.int8),
        ("precision", np.uint8),
        ("rice_order", np.uint8),
        ("status", np.int32),
        ("code_bits", np.uint64),
        ("subframe_bits", np.uint64),
        ("sum_quotients", np.uint64),
        ("rice_params", np.uint8, (256,)),
    ],
    align=True,
)
assert RECORD_DTYPE.itemsize == 352, RECORD_DTYPE.itemsize

FRAME_RESULT_DTYPE = np.dtype(
    [
        ("channel_assignment", np.uint8),
        ("kind", np.uint8, (2,)),
        ("role", np.uint8, (2,)),
        ("analysis_status", np.uint8),
        ("pad", np.uint8, (2,)),
        ("dc_offset", np.int32, (2,)),
        ("bits", np.uint64, (4,)),
        ("lpc", RECORD_DTYPE, (2,)),
    ],
    align=True,
)
assert FRAME_RESULT_DTYPE.itemsize == 8 + 8 + 32 + 704, FRAME_RESULT_DTYPE.itemsize

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _declare(L):
    f32p, f64p = C.POINTER(C.c_float), C.POINTER(C.c_double)
    i32p, u32p, u8p = C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint8)
    L.orc_window_weights.argtypes = [C.c_uint32, C.c_float, C.c_size_t, f32p]
    L.orc_fill_windowed_signal.argtypes = [i32p, f32p, C.c_size_t, f32p]
    L.orc_auto_correlation_f64.argtypes = [C.c_size_t, f32p, C.c_size_t, f64p]
    L.orc_auto_correlation_f32.argtypes = [C.c_size_t, f32p, C.c_size_t, f32p]
    L.orc_auto_correlation_canonical_f64.argtypes = [C.c_size_t, f32p, C.c_size_t, f64p]
    L.orc_auto_correlation_nightly_f64.argtypes = [C.c_size_t, f32p, C.c_size_t, f64p, C.c_size_t]
    L.orc_symmetric_levinson_f64.argtypes = [f64p, f64p, C.c_size_t, f64p]
    L.orc_symmetric_levinson_f64.restype = C.c_int
    L.orc_symmetric_levinson_f32.argtypes = [f32p, f32p, C.c_size_t, f32p]
    L.orc_symmetric_levinson_f32.restype = C.c_int
    L.orc_find_shift.argtypes = [f64p, C.c_size_t, C.c_uint32]
    L.orc_find_shift.restype = C.c_int32
    L.orc_quantize_parameters.argtypes = [f64p, C.c_size_t, C.c_uint32, C.POINTER(QParams)]
    L.orc_compute_error.argtypes = [C.POINTER(QParams), i32p, C.c_size_t, i32p]
    L.orc_lpc_from_autocorr.argtypes = [i32p, C.c_size_t, C.POINTER(QlpcConfig), f64p, f64p]
    L.orc_lpc_from_autocorr.restype = C.c_int
    L.orc_weighted_lagged_outer_prod_sum_f64.argtypes = [C.c_size_t, f32p, C.c_size_t, f32p, C.c_size_t, f64p]
    L.orc_weighted_lagged_outer_prod_sum_f64.restype = None
    L.orc_cholesky_solve.argtypes = [f64p, C.c_size_t, f64p]
    L.orc_cholesky_solve.restype = C.c_int
    L.orc_weighted_lpc_with_direct_mse.argtypes = [i32p, C.c_size_t, C.POINTER(QlpcConfig), f32p, f64p, f64p, f64p]
    L.orc_weighted_lpc_with_direct_mse.restype = C.c_int
    L.orc_lpc_with_irls_mae.argtypes = [i32p, C.c_size_t, C.POINTER(QlpcConfig), C.c_size_t, f64p, f64p]
    L.orc_lpc_with_irls_mae.restype = C.c_int
    L.orc_compute_raw_errors.argtypes = [i32p, C.c_size_t, f64p, C.c_size_t, f32p]
    L.orc_compute_raw_errors.restype = None
    L.orc_encode_signbit.argtypes = [C.c_int32]
    L.orc_encode_signbit.restype = C.c_uint32
    L.orc_decode_signbit.argtypes = [C.c_uint32]
    L.orc_decode_signbit.restype = C.c_int32
    L.orc_prc_bit_table_from_errors.argtypes = [u32p, C.c_size_t, C.c_uint32, u32p]
    L.orc_prc_minimizer.argtypes = [u32p, C.c_uint32, u32p, u32p]
    L.orc_prc_merge.argtypes = [u32p, u32p, C.c_uint32, u32p]
    L.orc_finest_partition_order.argtypes = [C.c_size_t, C.c_size_t]
    L.orc_finest_partition_order.restype = C.c_uint32
    L.orc_find_partitioned_rice_parameter.argtypes = [
        i32p, C.c_size_t, C.c_size_t, C.c_uint32, C.POINTER(PrcParameter)]
    L.orc_encode_residual_with_prc_parameter.argtypes = [
        i32p, C.c_size_t, C.c_size_t, C.POINTER(PrcParameter), u32p, u32p,
        C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.orc_residual_count_bits.argtypes = [
        C.c_size_t, C.c_size_t, C.c_uint32, u8p, C.c_uint64, C.c_uint64]
    L.orc_residual_count_bits.restype = C.c_uint64
    L.orc_lpc_count_bits.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64]
    L.orc_lpc_count_bits.restype = C.c_uint64
    L.orc_verbatim_count_bits.argtypes = [C.c_size_t, C.c_uint32]
    L.orc_verbatim_count_bits.restype = C.c_uint64
    L.orc_decode_residual.argtypes = [C.c_size_t, C.c_uint32, u8p, u32p, u32p, i32p]
    L.orc_decode_lpc.argtypes = [i32p, C.c_size_t, C.POINTER(C.c_int16), C.c_uint32, i32p,
                                 C.c_size_t, i32p]
    L.orc_estimated_qlpc.argtypes = [i32p, C.c_size_t, C.c_uint32, C.POINTER(QlpcConfig),
                                     C.POINTER(QlpcResult), u8p, i32p, u32p, u32p]
    L.orc_qlpc_batch.argtypes = [i32p, C.c_size_t, C.c_size_t, C.c_size_t, u8p,
                                 C.POINTER(QlpcConfig), C.c_void_p, i32p, C.c_size_t, f64p, f64p,
                                 C.c_int]
    L.orc_stereo_to_midside.argtypes = [i32p, i32p, C.c_size_t, i32p, i32p]
    L.orc_midside_to_stereo.argtypes = [i32p, i32p, C.c_size_t, i32p, i32p]
    L.orc_is_constant.argtypes = [i32p, C.c_size_t]
    L.orc_is_constant.restype = C.c_int
    L.orc_bench_qlpc.argtypes = [i32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint32,
                                 C.POINTER(QlpcConfig), C.c_int, C.c_int]
    L.orc_bench_qlpc.restype = C.c_double
    L.orc_bench_stereo_qlpc.argtypes = [i32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint32,
                                        C.POINTER(QlpcConfig), C.c_int, C.c_int,
                                        C.POINTER(C.c_uint64)]
    L.orc_bench_stereo_qlpc.restype = C.c_double
    L.orc_encode_stereo_frame.argtypes = [i32p, i32p, C.c_size_t, C.c_uint32, C.POINTER(QlpcConfig),
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                          i32p, i32p]


    L.orc_log2f.argtypes = [C.c_float]
    L.orc_log2f.restype = C.c_float
    L.orc_reset_fixed_lpc_errors.argtypes = [i32p, C.c_size_t, i32p]
    L.orc_find_sum_abs_f32.argtypes = [i32p, C.c_size_t, C.c_int, C.c_size_t]
    L.orc_find_sum_abs_f32.restype = C.c_float
    L.orc_estimate_entropy.argtypes = [i32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
    L.orc_estimate_entropy.restype = C.c_uint64
    L.orc_fixed_lpc.argtypes = [i32p, C.c_size_t, C.c_uint32, C.c_uint64, C.POINTER(FixedConfig),
                                C.c_uint32, C.POINTER(FixedResult), u8p, i32p]
    L.orc_fixed_lpc.restype = C.c_int
    L.orc_encode_subframe.argtypes = [i32p, C.c_size_t, C.c_uint32, C.POINTER(FrameConfig),
                                      C.POINTER(C.c_uint64), C.POINTER(QlpcResult),
                                      C.POINTER(FixedResult), u8p, i32p]
    L.orc_encode_subframe.restype = C.c_int
    L.orc_encode_stereo_frame_cfg.argtypes = [i32p, i32p, C.c_size_t, C.c_uint32,
                                              C.POINTER(FrameConfig), C.c_void_p, i32p, i32p]


    L.orc_crc8.argtypes = [u8p, C.c_size_t]
    L.orc_crc8.restype = C.c_uint8
    L.orc_crc16.argtypes = [u8p, C.c_size_t]
    L.orc_crc16.restype = C.c_uint16
    L.orc_encode_to_utf8like.argtypes = [C.c_uint64, u8p]
    L.orc_encode_to_utf8like.restype = C.c_size_t
    L.orc_write_frame_header.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint64, u8p]
    L.orc_write_frame_header.restype = C.c_size_t
    L.orc_write_subframe.argtypes = [C.POINTER(SubframeDesc), C.c_size_t, u8p, C.c_size_t]
    L.orc_write_subframe.restype = C.c_size_t
    L.orc_write_frame.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.POINTER(SubframeDesc), u8p, C.c_size_t]
    L.orc_write_frame.restype = C.c_size_t
    L.orc_write_stereo_frame.argtypes = [C.c_void_p, i32p, i32p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                         i32p, i32p, u8p, C.c_size_t]
    L.orc_write_stereo_frame.restype = C.c_size_t


    L.orc_le_bytes_to_i32s.argtypes = [u8p, C.c_size_t, i32p, C.c_uint32]
    L.orc_deinterleave.argtypes = [i32p, C.c_size_t, C.c_size_t, C.c_size_t, i32p]


def make_fixed_config(max_order=4, order_sel=ORDERSEL_APPROXENT, partitions=16,
                      sum_mode=SUMABS_STABLE) -> FixedConfig:
    """config::Fixed defaults: max_order 4 (constant.rs:95), ApproxEnt{16} (config.rs:411-417)."""
    return FixedConfig(max_order, order_sel, partitions, sum_mode)


def make_frame_config(qlpc=None, use_constant=True, use_fixed=True, use_lpc=True, use_leftside=True,
                      use_rightside=True, use_midside=True, fixed=None) -> FrameConfig:
    """config::SubFrameCoding / StereoCoding defaults: everything on (config.rs:185-196, 146-154)."""
    return FrameConfig(qlpc or make_config(), int(use_constant), int(use_fixed), int(use_lpc),
                       int(use_leftside), int(use_rightside), int(use_midside),
                       fixed or make_fixed_config())


RICE_FINEST_ONLY = 0x100


def make_config(lpc_order=10, quant_precision=15, window=("tukey", 0.4), max_rice_parameter=30,
                acorr=ACORR_REFERENCE, rice_finest_only=False, use_direct_mse=False,
                mae_optimization_steps=0) -> QlpcConfig:
    """config::Qlpc / config::Prc defaults, src/constant.rs:109-115, src/config.rs:216-221.
    use_direct_mse / mae_optimization_steps: the experimental estimators of src/coding.rs:337-347."""
    if window == "rectangle" or window[0] == "rectangle":
        wt, alpha = WINDOW_RECTANGLE, 0.0
    else:
        wt, alpha = WINDOW_TUKEY, float(window[1])
    if use_direct_mse:
        acorr = ACORR_DIRECT_MSE | (int(mae_optimization_steps) << 8)
    return QlpcConfig(lpc_order, quant_precision, wt, alpha,
                      max_rice_parameter | (RICE_FINEST_ONLY if rice_finest_only else 0), acorr)


# ---------------------------------------------------------------- lpc.rs ----
def default_order_is_certified(n: int, lpc_order: int) -> bool:
    """Shapes on which the unflagged product certifies its chunk-tree sums against the reference's chains."""
    L = lib()
    L.orc_default_order_is_certified.argtypes = [C.c_size_t, C.c_size_t]
    L.orc_default_order_is_certified.restype = C.c_int
    return bool(L.orc_default_order_is_certified(n, lpc_order))


def certificate_bounds(signal, cfg: "QlpcConfig"):
    """The order certificate's quantities for one subframe of a certified shape (orc_certificate_bounds): R[] in the fused
    kernel's lane order, the recursion's coefficients, the second tier's per-coefficient bound da (safety included) and the
    first tier's uniform bound.  None where the certificate does not apply (silence, a skipped step, another shape)."""
    x = np.ascontiguousarray(signal, np.int32)
    P = int(cfg.lpc_order)
    corr = np.zeros(33, np.float64)
    coefs = np.zeros(32, np.float64)
    da = np.zeros(32, np.float64)
    tier1 = C.c_double(0.0)
    L = lib()
    L.orc_certificate_bounds.restype = C.c_int
    rc = L.orc_certificate_bounds(_p(x, C.c_int32), C.c_size_t(len(x)), C.byref(cfg), _p(corr, C.c_double),
                                  _p(coefs, C.c_double), _p(da, C.c_double), C.byref(tier1))
    if rc != 0:
        return None
    return {"R": corr[: P + 1].copy(), "a": coefs[:P].copy(), "da": da[:P].copy(), "tier1": float(tier1.value)}


def cert_stats(reset: bool = False):
    """(subframes analysed in the certified mode, certificates that needed the rows of T^-1, subframes recomputed in the
    reference's order) since the last reset.  Plain counters: read them after single-threaded runs."""
    arr = (C.c_ulong * 3).in_dll(lib(), "orc_cert_stats")
    out = (int(arr[0]), int(arr[1]), int(arr[2]))
    if reset:
        arr[0] = arr[1] = arr[2] = 0
    return out


def window_weights(window, n: int) -> np.ndarray:
    cfg = make_config(window=window)
    out = np.empty(n, np.float32)
    lib().orc_window_weights(cfg.window_type, cfg.tukey_alpha, n, _p(out, C.c_float))
    return out


def fill_windowed_signal(signal, window) -> np.ndarray:
    s = np.ascontiguousarray(signal, np.int32)
    w = np.ascontiguousarray(window, np.float32)
    out = np.empty(len(s), np.float32)
    lib().orc_fill_windowed_signal(_p(s, C.c_int32), _p(w, C.c_float), len(s), _p(out, C.c_float))
    return out


def auto_correlation_nightly(order: int, signal, base_mod: int = 0) -> np.ndarray:
    """simd-nightly summation order (src/lpc.rs:510-531, 439-500)."""
    x = np.ascontiguousarray(signal, np.float32)
    out = np.zeros(order, np.float64)
    lib().orc_auto_correlation_nightly_f64(order, _p(x, C.c_float), len(x), _p(out, C.c_double), base_mod)
    return out


def auto_correlation(order: int, signal, dtype=np.float64, canonical=False) -> np.ndarray:
    x = np.ascontiguousarray(signal, np.float32)
    if dtype == np.float32:
        out = np.zeros(order, np.float32)
        lib().orc_auto_correlation_f32(order, _p(x, C.c_float), len(x), _p(out, C.c_float))
        return out
    out = np.zeros(order, np.float64)
    fn = lib().orc_auto_correlation_canonical_f64 if canonical else lib().orc_auto_correlation_f64
    fn(order, _p(x, C.c_float), len(x), _p(out, C.c_double))
    return out


def symmetric_levinson_recursion(coefs, ys, dtype=np.float64):
    if dtype == np.float32:
        c = np.ascontiguousarray(coefs, np.float32)
        y = np.ascontiguousarray(ys, np.float32)
        out = np.zeros(len(y), np.float32)
        st = lib().orc_symmetric_levinson_f32(_p(c, C.c_float), _p(y, C.c_float), len(y),
                                              _p(out, C.c_float))
    else:
        c = np.ascontiguousarray(coefs, np.float64)
        y = np.ascontiguousarray(ys, np.float64)
        out = np.zeros(len(y), np.float64)
        st = lib().orc_symmetric_levinson_f64(_p(c, C.c_double), _p(y, C.c_double), len(y),
                                              _p(out, C.c_double))
    return out, st


def find_shift(coefs, precision: int) -> int:
    c = np.ascontiguousarray(coefs, np.float64)
    return int(lib().orc_find_shift(_p(c, C.c_double), len(c), precision))


def quantize_parameters(coefs, precision: int) -> QParams:
    c = np.ascontiguousarray(coefs, np.float64)
    qp = QParams()
    lib().orc_quantize_parameters(_p(c, C.c_double), len(c), precision, C.byref(qp))
    return qp


def qparams(coefs, shift: int, precision: int) -> QParams:
    qp = QParams()
    for i, v in enumerate(coefs):
        qp.coefs[i] = int(v)
    qp.order, qp.shift, qp.precision = len(coefs), shift, precision
    return qp


def compute_error(qp: QParams, signal) -> np.ndarray:
    s = np.ascontiguousarray(signal, np.int32)
    out = np.zeros(len(s), np.int32)
    lib().orc_compute_error(C.byref(qp), _p(s, C.c_int32), len(s), _p(out, C.c_int32))
    return out


def lpc_from_autocorr(signal, cfg: QlpcConfig):
    s = np.ascontiguousarray(signal, np.int32)
    corr = np.zeros(33, np.float64)
    coefs = np.zeros(32, np.float64)
    st = lib().orc_lpc_from_autocorr(_p(s, C.c_int32), len(s), C.byref(cfg), _p(corr, C.c_double),
                                     _p(coefs, C.c_double))
    return corr[: cfg.lpc_order + 1], coefs[: cfg.lpc_order], st


# ---- experimental: covariance-method LPC and IRLS (src/lpc.rs:573-618, 814-903; parity unpinned) ----
def lagged_outer_prod_sum(order: int, signal, weight=None, wshift: int = 0) -> np.ndarray:
    """weighted_lagged_outer_prod_sum (src/lpc.rs:573-600) -> [order, order] f64"""
    x = np.ascontiguousarray(signal, np.float32)
    out = np.zeros(order * order, np.float64)
    w = None if weight is None else np.ascontiguousarray(weight, np.float32)
    lib().orc_weighted_lagged_outer_prod_sum_f64(order, _p(x, C.c_float), len(x),
                                                 None if w is None else _p(w, C.c_float), wshift, _p(out, C.c_double))
    return out.reshape(order, order).T.copy()  # column-major -> [row, col]


def cholesky_solve(mat, v):
    """LpcFloat::solve_sym_mut (src/lpc.rs:79-87; nalgebra's Cholesky): (ok, solution)"""
    m = np.asfortranarray(np.asarray(mat, np.float64))
    n = m.shape[0]
    flat = np.ascontiguousarray(m.T).reshape(-1)  # column-major buffer
    x = np.ascontiguousarray(v, np.float64).copy()
    ok = lib().orc_cholesky_solve(_p(flat, C.c_double), n, _p(x, C.c_double))
    return bool(ok), x


def lpc_with_direct_mse(signal, cfg: QlpcConfig, weight=None):
    """LpcEstimator::weighted_lpc_with_direct_mse (src/lpc.rs:853-903) -> (autocorr, gram, coefs, status)"""
    s = np.ascontiguousarray(signal, np.int32)
    P = cfg.lpc_order
    corr, gram, coefs = np.zeros(33, np.float64), np.zeros(32 * 32, np.float64), np.zeros(32, np.float64)
    w = None if weight is None else np.ascontiguousarray(weight, np.float32)
    st = lib().orc_weighted_lpc_with_direct_mse(_p(s, C.c_int32), len(s), C.byref(cfg),
                                                None if w is None else _p(w, C.c_float), _p(corr, C.c_double),
                                                _p(gram, C.c_double), _p(coefs, C.c_double))
    return corr[: P + 1], gram[: P * P].reshape(P, P).T.copy(), coefs[:P], st


def lpc_with_irls_mae(signal, cfg: QlpcConfig, steps: int):
    """LpcEstimator::lpc_with_irls_mae (src/lpc.rs:814-850) -> (coefs, status)"""
    s = np.ascontiguousarray(signal, np.int32)
    corr, coefs = np.zeros(33, np.float64), np.zeros(32, np.float64)
    st = lib().orc_lpc_with_irls_mae(_p(s, C.c_int32), len(s), C.byref(cfg), steps, _p(corr, C.c_double),
                                     _p(coefs, C.c_double))
    return coefs[: cfg.lpc_order], st


def compute_raw_errors(signal, coefs) -> np.ndarray:
    """compute_raw_errors (src/lpc.rs:602-618): prediction - signal in f32, zeros in the warm-up"""
    s = np.ascontiguousarray(signal, np.int32)
    c = np.ascontiguousarray(coefs, np.float64)
    out = np.zeros(len(s), np.float32)
    lib().orc_compute_raw_errors(_p(s, C.c_int32), len(s), _p(c, C.c_double), len(c), _p(out, C.c_float))
    return out


# --------------------------------------------------------------- rice.rs ----
def encode_signbit(v: int) -> int:
    return int(lib().orc_encode_signbit(int(v)))


def decode_signbit(v: int) -> int:
    return int(lib().orc_decode_signbit(int(v)))


def prc_bit_table_from_errors(errors, offset: int) -> np.ndarray:
    e = np.ascontiguousarray(errors, np.uint32)
    out = np.zeros(32, np.uint32)
    lib().orc_prc_bit_table_from_errors(_p(e, C.c_uint32), len(e), offset, _p(out, C.c_uint32))
    return out


def prc_minimizer(table, max_p: int):
    t = np.ascontiguousarray(table, np.uint32)
    p, bits = C.c_uint32(), C.c_uint32()
    lib().orc_prc_minimizer(_p(t, C.c_uint32), max_p, C.byref(p), C.byref(bits))
    return p.value, bits.value


def prc_merge(a, b, offset: int) -> np.ndarray:
    a = np.ascontiguousarray(a, np.uint32)
    b = np.ascontiguousarray(b, np.uint32)
    out = np.zeros(32, np.uint32)
    lib().orc_prc_merge(_p(a, C.c_uint32), _p(b, C.c_uint32), offset, _p(out, C.c_uint32))
    return out


def finest_partition_order(size: int, min_part_size: int) -> int:
    return int(lib().orc_finest_partition_order(size, min_part_size))


def find_partitioned_rice_parameter(signal, warmup_length: int, max_p: int):
    s = np.ascontiguousarray(signal, np.int32)
    prc = PrcParameter()
    lib().orc_find_partitioned_rice_parameter(_p(s, C.c_int32), len(s), warmup_length, max_p,
                                              C.byref(prc))
    ps = np.frombuffer(prc.ps, np.uint8, 1 << prc.order).copy()
    return prc.order, ps, int(prc.code_bits), prc


def encode_residual(errors, warmup_length: int, max_p: int = 30):
    """coding::encode_residual (src/coding.rs:173-176) -> dict mirroring component::Residual."""
    e = np.ascontiguousarray(errors, np.int32)
    order, ps, code_bits, prc = find_partitioned_rice_parameter(e, warmup_length, max_p)
    q = np.zeros(len(e), np.uint32)
    r = np.zeros(len(e), np.uint32)
    sq, sp = C.c_uint64(), C.c_uint64()
    lib().orc_encode_residual_with_prc_parameter(_p(e, C.c_int32), len(e), warmup_length,
                                                 C.byref(prc), _p(q, C.c_uint32),
                                                 _p(r, C.c_uint32), C.byref(sq), C.byref(sp))
    bits = residual_count_bits(len(e), warmup_length, order, ps, sq.value, sp.value)
    return dict(partition_order=order, rice_params=ps, quotients=q, remainders=r,
                sum_quotients=sq.value, sum_rice_params=sp.value, code_bits=code_bits,
                count_bits=bits, block_size=len(e), warmup_length=warmup_length)


def residual_count_bits(block_size, warmup_length, partition_order, rice_params, sum_quotients,
                        sum_rice_params) -> int:
    rp = np.ascontiguousarray(rice_params, np.uint8)
    return int(lib().orc_residual_count_bits(block_size, warmup_length, partition_order,
                                             _p(rp, C.c_uint8), sum_quotients, sum_rice_params))


def lpc_count_bits(bits_per_sample, order, precision, residual_bits) -> int:
    return int(lib().orc_lpc_count_bits(bits_per_sample, order, precision, residual_bits))


def verbatim_count_bits(n, bits_per_sample) -> int:
    return int(lib().orc_verbatim_count_bits(n, bits_per_sample))


def decode_residual(res: dict) -> np.ndarray:
    out = np.zeros(res["block_size"], np.int32)
    rp = np.ascontiguousarray(res["rice_params"], np.uint8)
    lib().orc_decode_residual(res["block_size"], res["partition_order"], _p(rp, C.c_uint8),
                              _p(res["quotients"], C.c_uint32), _p(res["remainders"], C.c_uint32),
                              _p(out, C.c_int32))
    return out


def decode_lpc(warm_up, coefs, shift: int, residual) -> np.ndarray:
    w = np.ascontiguousarray(warm_up, np.int32)
    c = np.ascontiguousarray(coefs, np.int16)
    r = np.ascontiguousarray(residual, np.int32)
    out = np.zeros(len(r), np.int32)
    lib().orc_decode_lpc(_p(w, C.c_int32), len(w), _p(c, C.c_int16), shift, _p(r, C.c_int32),
                         len(r), _p(out, C.c_int32))
    return out


# ------------------------------------------------------------- coding.rs ----
def estimated_qlpc(signal, bits_per_sample: int, cfg: QlpcConfig) -> dict:
    """coding::estimated_qlpc (src/coding.rs:360-381) for one subframe."""
    s = np.ascontiguousarray(signal, np.int32)
    n = len(s)
    res = QlpcResult()
    rp = np.zeros(32768, np.uint8)
    err = np.zeros(n, np.int32)
    q = np.zeros(n, np.uint32)
    r = np.zeros(n, np.uint32)
    lib().orc_estimated_qlpc(_p(s, C.c_int32), n, bits_per_sample, C.byref(cfg), C.byref(res),
                             _p(rp, C.c_uint8), _p(err, C.c_int32), _p(q, C.c_uint32),
                             _p(r, C.c_uint32))
    order = res.qp.order
    return dict(
        status=res.status,
        coefs=np.array(res.qp.coefs[:order], np.int16),
        order=order, shift=res.qp.shift, precision=res.qp.precision,
        warm_up=s[:order].copy(), residual=err,
        rice_order=res.rice_order, rice_params=rp[: 1 << res.rice_order].copy(),
        quotients=q, remainders=r,
        code_bits=int(res.code_bits), sum_quotients=int(res.sum_quotients),
        residual_bits=int(res.residual_bits), subframe_bits=int(res.subframe_bits),
        autocorr=np.array(res.autocorr[: cfg.lpc_order + 1]),
        lpc_coefs=np.array(res.lpc_coefs[: cfg.lpc_order]),
    )


def qlpc_batch(samples, bps, cfg: QlpcConfig, nthreads: int = 1, want_fp: bool = True):
    """estimated_qlpc over a [n_subframes, n] int32 array.  Returns (records, residual, R, a)."""
    x = np.ascontiguousarray(samples, np.int32)
    ns, n = x.shape
    recs = np.zeros(ns, RECORD_DTYPE)
    residual = np.zeros((ns, n), np.int32)
    bps_a = np.ascontiguousarray(np.broadcast_to(np.asarray(bps, np.uint8), (ns,)))
    R = np.zeros((ns, 33), np.float64) if want_fp else None
    A = np.zeros((ns, 32), np.float64) if want_fp else None
    lib().orc_qlpc_batch(_p(x, C.c_int32), ns, n, n, _p(bps_a, C.c_uint8), C.byref(cfg),
                         recs.ctypes.data_as(C.c_void_p), _p(residual, C.c_int32), n,
                         _p(R, C.c_double) if want_fp else None,
                         _p(A, C.c_double) if want_fp else None, nthreads)
    return recs, residual, R, A


def stereo_to_midside(l, r):
    l = np.ascontiguousarray(l, np.int32)
    r = np.ascontiguousarray(r, np.int32)
    m = np.zeros_like(l)
    s = np.zeros_like(l)
    lib().orc_stereo_to_midside(_p(l, C.c_int32), _p(r, C.c_int32), len(l), _p(m, C.c_int32),
                                _p(s, C.c_int32))
    return m, s


def midside_to_stereo(m, s):
    m = np.ascontiguousarray(m, np.int32)
    s = np.ascontiguousarray(s, np.int32)
    l = np.zeros_like(m)
    r = np.zeros_like(m)
    lib().orc_midside_to_stereo(_p(m, C.c_int32), _p(s, C.c_int32), len(m), _p(l, C.c_int32),
                                _p(r, C.c_int32))
    return l, r


def is_constant(samples) -> bool:
    s = np.ascontiguousarray(samples, np.int32)
    return bool(lib().orc_is_constant(_p(s, C.c_int32), len(s)))


# ------------------------------------------------------ fixed LPC ----
def log2f(x: float) -> float:
    return float(lib().orc_log2f(float(np.float32(x))))


def reset_fixed_lpc_errors(signal) -> np.ndarray:
    """reset_fixed_lpc_errors, src/coding.rs:182-197 -> int32 [5, n]."""
    x = np.ascontiguousarray(signal, np.int32)
    out = np.zeros((5, len(x)), np.int32)
    lib().orc_reset_fixed_lpc_errors(_p(x, C.c_int32), len(x), _p(out, C.c_int32))
    return out


def find_sum_abs_f32(data, mode=SUMABS_STABLE, base_mod=0) -> float:
    x = np.ascontiguousarray(data, np.int32)
    return float(lib().orc_find_sum_abs_f32(_p(x, C.c_int32), len(x), mode, base_mod))


def estimate_entropy(errors, warmup_len: int, partitions: int, mode=SUMABS_STABLE) -> int:
    x = np.ascontiguousarray(errors, np.int32)
    return int(lib().orc_estimate_entropy(_p(x, C.c_int32), len(x), warmup_len, partitions, mode))


def fixed_lpc(signal, bits_per_sample: int, baseline_bits: int, fixed: FixedConfig = None,
              max_p: int = 30):
    """fixed_lpc, src/coding.rs:298-331 -> dict or None (the Option)."""
    x = np.ascontiguousarray(signal, np.int32)
    n = len(x)
    fc = fixed or make_fixed_config()
    res = FixedResult()
    rp = np.zeros(32768, np.uint8)
    err = np.zeros(n, np.int32)
    sel = lib().orc_fixed_lpc(_p(x, C.c_int32), n, bits_per_sample, min(baseline_bits, 2 ** 64 - 1),
                              C.byref(fc), max_p, C.byref(res), _p(rp, C.c_uint8), _p(err, C.c_int32))
    out = {"selected": bool(sel), "order": int(res.order), "estimate": list(res.estimate)}
    if sel:
        out.update(rice_order=int(res.rice_order), code_bits=int(res.code_bits),
                   sum_quotients=int(res.sum_quotients), residual_bits=int(res.residual_bits),
                   subframe_bits=int(res.subframe_bits), rice_params=rp[:1 << res.rice_order].copy(),
                   residual=err, warm_up=x[:res.order].copy())
    return out


def decode_fixed(warm_up, residual) -> np.ndarray:
    """Decode for FixedLpc, src/component/decode.rs:187-201."""
    order = len(warm_up)
    return decode_lpc(warm_up, np.array(FIXED_LPC_COEFS[order][:order], np.int16), 0, residual)


def encode_subframe(samples, bits_per_sample: int, fc: FrameConfig) -> dict:
    """encode_subframe, src/coding.rs:384-418."""
    x = np.ascontiguousarray(samples, np.int32)
    n = len(x)
    bits = C.c_uint64()
    lpc, fixed = QlpcResult(), FixedResult()
    rp = np.zeros(32768, np.uint8)
    err = np.zeros(n, np.int32)
    kind = lib().orc_encode_subframe(_p(x, C.c_int32), n, bits_per_sample, C.byref(fc), C.byref(bits),
                                     C.byref(lpc), C.byref(fixed), _p(rp, C.c_uint8), _p(err, C.c_int32))
    return {"kind": kind, "bits": int(bits.value), "lpc": lpc, "fixed": fixed, "rice_params": rp,
            "residual": err}


def encode_stereo_frames_cfg(frames, bps: int, fc: FrameConfig):
    """encode_frame for 2-channel frames (src/coding.rs:530-544, 469-527), every candidate.

    `frames` int32 [n_frames, 2, n] -> (results FRAME_RESULT_DTYPE [n_frames], residual [n_frames, 2, n])."""
    x = np.ascontiguousarray(frames, np.int32)
    nf, ch, n = x.shape
    assert ch == 2
    out = np.zeros(nf, FRAME_RESULT_DTYPE)
    resid = np.zeros((nf, 2, n), np.int32)
    for f in range(nf):
        lib().orc_encode_stereo_frame_cfg(_p(x[f, 0], C.c_int32), _p(x[f, 1], C.c_int32), n, bps,
                                          C.byref(fc), out[f:f + 1].ctypes.data_as(C.c_void_p),
                                          _p(resid[f, 0], C.c_int32), _p(resid[f, 1], C.c_int32))
    return out, resid


# ------------------------------------------------------ bit writer ----
def crc8(data: bytes) -> int:
    b = np.frombuffer(bytes(data), np.uint8).copy()
    return int(lib().orc_crc8(_p(b, C.c_uint8), len(b)))


def crc16(data: bytes) -> int:
    b = np.frombuffer(bytes(data), np.uint8).copy()
    return int(lib().orc_crc16(_p(b, C.c_uint8), len(b)))


def encode_to_utf8like(val: int):
    out = np.zeros(7, np.uint8)
    n = lib().orc_encode_to_utf8like(val, _p(out, C.c_uint8))
    return bytes(out[:n]) if n else None


def write_frame_header(block_size, channel_tag, bits_per_sample, sample_rate, variable, offset) -> bytes:
    out = np.zeros(24, np.uint8)
    n = lib().orc_write_frame_header(block_size, channel_tag, bits_per_sample, sample_rate, int(variable), offset,
                                     _p(out, C.c_uint8))
    return bytes(out[:n])


def _desc(kind, bps, n, samples=None, dc_offset=0, order=0, shift=0, precision=0, coefs=None, rice_order=0,
          rice_params=None, residual=None):
    keep = [np.ascontiguousarray(samples if samples is not None else np.zeros(n), np.int32),
            np.ascontiguousarray(coefs if coefs is not None else np.zeros(32), np.int16),
            np.ascontiguousarray(rice_params if rice_params is not None else np.zeros(1), np.uint8),
            np.ascontiguousarray(residual if residual is not None else np.zeros(n), np.int32)]
    d = SubframeDesc(kind, bps, dc_offset, _p(keep[0], C.c_int32), order, shift, precision, _p(keep[1], C.c_int16),
                     rice_order, _p(keep[2], C.c_uint8), _p(keep[3], C.c_int32))
    return d, keep


def write_subframe(kind, bps, n, **kw):
    """BitRepr for SubFrame::write (bitrepr.rs:449-527) -> (bytes, bit count)."""
    d, keep = _desc(kind, bps, n, **kw)
    cap = n * 5 + 64
    out = np.zeros(cap, np.uint8)
    bits = lib().orc_write_subframe(C.byref(d), n, _p(out, C.c_uint8), cap)
    return bytes(out[:(bits + 7) // 8]), int(bits)


def write_frame(block_size, channel_assignment, bits_per_sample, sample_rate, frame_number, subframes) -> bytes:
    """Frame::write (bitrepr.rs:289-319); `subframes` = list of dicts with the _desc keywords
    (kind, bps, samples, dc_offset, order, shift, precision, coefs, rice_order, rice_params, residual)."""
    keep, descs = [], (SubframeDesc * len(subframes))()
    for i, sf in enumerate(subframes):
        d, k = _desc(n=block_size, **sf)
        descs[i] = d
        keep.append(k)
    cap = len(subframes) * (block_size * 4 + 16) + 64
    out = np.zeros(cap, np.uint8)
    ln = lib().orc_write_frame(block_size, channel_assignment, len(subframes), bits_per_sample, sample_rate,
                               frame_number, descs, _p(out, C.c_uint8), cap)
    assert ln <= cap
    return bytes(out[:ln])


def write_stereo_frame(result, l, r, bps, sample_rate, frame_number, residual0, residual1) -> bytes:
    """Frame::write (bitrepr.rs:289-319) of the frame one FRAME_RESULT_DTYPE record describes."""
    rec = np.ascontiguousarray(result.reshape(1) if hasattr(result, "reshape") else np.array([result]))
    l = np.ascontiguousarray(l, np.int32)
    r = np.ascontiguousarray(r, np.int32)
    r0 = np.ascontiguousarray(residual0, np.int32)
    r1 = np.ascontiguousarray(residual1, np.int32)
    n = len(l)
    cap = n * 8 + 64
    out = np.zeros(cap, np.uint8)
    ln = lib().orc_write_stereo_frame(rec.ctypes.data_as(C.c_void_p), _p(l, C.c_int32), _p(r, C.c_int32), n, bps,
                                      sample_rate, frame_number, _p(r0, C.c_int32), _p(r1, C.c_int32),
                                      _p(out, C.c_uint8), cap)
    assert ln <= cap
    return bytes(out[:ln])


# ------------------------------------------------------ input side ----
def le_bytes_to_i32s(data: bytes, bytes_per_sample: int) -> np.ndarray:
    """le_bytes_to_i32s, src/arrayutils.rs:273-290."""
    b = np.frombuffer(bytes(data), np.uint8).copy()
    out = np.zeros(len(b) // bytes_per_sample, np.int32)
    lib().orc_le_bytes_to_i32s(_p(b, C.c_uint8), len(b), _p(out, C.c_int32), bytes_per_sample)
    return out


def deinterleave(interleaved, channels: int, channel_stride: int, dest_len: int = None, fill: int = 0) -> np.ndarray:
    """deinterleave, src/arrayutils.rs:248-264 -> flat dest (channels * channel_stride, or `dest_len`)."""
    x = np.ascontiguousarray(interleaved, np.int32)
    out = np.full(dest_len or channels * channel_stride, fill, np.int32)
    lib().orc_deinterleave(_p(x, C.c_int32), len(x), channels, channel_stride, _p(out, C.c_int32))
    return out


def bench_qlpc(samples, bits_per_sample: int, cfg: QlpcConfig, nthreads: int, repeats: int = 1):
    """Wall seconds for `repeats` passes of estimated_qlpc over the batch (cpu_baseline)."""
    x = np.ascontiguousarray(samples, np.int32)
    ns, n = x.shape
    return float(lib().orc_bench_qlpc(_p(x, C.c_int32), ns, n, n, bits_per_sample, C.byref(cfg),
                                      nthreads, repeats))


def bench_stereo_qlpc(frames, bits_per_sample: int, cfg: QlpcConfig, nthreads: int, repeats: int = 1):
    """Wall seconds for `repeats` passes of the stereo workload (L, R, M, S through the path per
    frame) over int32 [n_frames, 2, n]; also returns the sum of subframe_bits as a checksum."""
    x = np.ascontiguousarray(frames, np.int32)
    nf, ch, n = x.shape
    assert ch == 2
    chk = C.c_uint64()
    secs = float(lib().orc_bench_stereo_qlpc(_p(x, C.c_int32), nf, n, n, bits_per_sample,
                                             C.byref(cfg), nthreads, repeats, C.byref(chk)))
    return secs, int(chk.value)


def encode_stereo_frames(frames, bps: int, cfg: QlpcConfig, use_constant=True, use_lpc=True,
                         use_leftside=True, use_rightside=True, use_midside=True):
    """encode_frame for 2-channel frames (src/coding.rs:530-544, 469-527) with use_fixed = false.

    `frames` int32 [n_frames, 2, n] -> (results FRAME_RESULT_DTYPE [n_frames], residual [n_frames, 2, n])."""
    x = np.ascontiguousarray(frames, np.int32)
    nf, ch, n = x.shape
    assert ch == 2
    out = np.zeros(nf, FRAME_RESULT_DTYPE)
    resid = np.zeros((nf, 2, n), np.int32)
    for f in range(nf):
        lib().orc_encode_stereo_frame(_p(x[f, 0], C.c_int32), _p(x[f, 1], C.c_int32), n, bps, C.byref(cfg),
                                      int(use_constant), int(use_lpc), int(use_leftside),
                                      int(use_rightside), int(use_midside),
                                      out[f:f + 1].ctypes.data_as(C.c_void_p),
                                      _p(resid[f, 0], C.c_int32), _p(resid[f, 1], C.c_int32))
    return out, resid
