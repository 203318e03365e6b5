"""ctypes binding of the C ABI in ``include/flacenc_hip.h`` (libflacenc_hip.so).

This is plumbing for the Python test/bench drivers: every compute call goes
through the same ``extern "C"`` entry points a Rust/C host would bind.  There is
no CPU fallback -- if the HIP library is missing or no GPU is present the calls
raise.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FLACENC_HIP_LIB: development aid for A/B timing of two builds on one GPU box; never a fallback
LIB_PATH = os.environ.get("FLACENC_HIP_LIB") or os.path.join(_HERE, "libflacenc_hip.so")
# The same objects + the test / profiling hooks of csrc/flacenc_hip_debug.h (the product library has none of them):
# what Handle(dev, hooks=True) loads.  An A/B library named by FLACENC_HIP_LIB (tools/variant_*.sh build them with the
# hooks) serves both.
HOOKS_LIB_PATH = os.environ.get("FLACENC_HIP_LIB") or os.path.join(_HERE, "libflacenc_hip_hooks.so")

OK = 0
ERR_BAD_CONFIG = -1
ERR_BAD_ARGUMENT = -2
ERR_DEVICE = -3
ERR_UNSUPPORTED = -4
ERR_NO_DEVICE = -5
_ERR_NAMES = {
    ERR_BAD_CONFIG: "BAD_CONFIG",
    ERR_BAD_ARGUMENT: "BAD_ARGUMENT",
    ERR_DEVICE: "DEVICE",
    ERR_UNSUPPORTED: "UNSUPPORTED",
    ERR_NO_DEVICE: "NO_DEVICE",
}

WINDOW_RECTANGLE = 0
WINDOW_TUKEY = 1
FLAG_ALLOW_ORDER_32 = 1
MEM_HOST = 0
MEM_DEVICE = 1
COMM_ID_BYTES = 128  # FLACENC_HIP_COMM_ID_BYTES

# every symbol include/flacenc_hip.h declares
ABI_VERSION = 6  # FLACENC_HIP_ABI_VERSION of include/flacenc_hip.h
DEBUG_SYMBOLS = ("flacenc_hip_debug_set_stamps", "flacenc_hip_debug_set_fixed_keys", "flacenc_hip_debug_set_cert_stats",
                 "flacenc_hip_debug_set_adaptive_order", "flacenc_hip_debug_adaptive_state")
EXPORTED_SYMBOLS = (
    "flacenc_hip_abi_version",
    "flacenc_hip_device_count",
    "flacenc_hip_create",
    "flacenc_hip_destroy",
    "flacenc_hip_last_error",
    "flacenc_hip_verify_config",
    "flacenc_hip_window_weights",
    "flacenc_hip_qlpc_batch",
    "flacenc_hip_qlpc_batch_async",
    "flacenc_hip_stereo_qlpc_batch",
    "flacenc_hip_stereo_qlpc_batch_async",
    "flacenc_hip_encode_stereo_frames",
    "flacenc_hip_encode_stereo_frames_async",
    "flacenc_hip_fixed_lpc_batch",
    "flacenc_hip_fixed_lpc_batch_async",
    "flacenc_hip_stereo_frame_bytes_bound",
    "flacenc_hip_pack_stereo_frames",
    "flacenc_hip_pack_stereo_frames_async",
    "flacenc_hip_stereo_frame_lengths_async",
    "flacenc_hip_place_frames_async",
    "flacenc_hip_frame_wire_bytes",
    "flacenc_hip_stereo_frame_wire_async",
    "flacenc_hip_stream_offsets_async",
    "flacenc_hip_encode_pcm_stereo",
    "flacenc_hip_encode_pcm",
    "flacenc_hip_host_alloc",
    "flacenc_hip_host_free",
    "flacenc_hip_set_host_threads",
    "flacenc_hip_encode_frames",
    "flacenc_hip_encode_frames_async",
    "flacenc_hip_frame_bytes_bound",
    "flacenc_hip_pack_frames",
    "flacenc_hip_pack_frames_async",
    "flacenc_hip_fill_le_bytes",
    "flacenc_hip_fill_le_bytes_async",
    "flacenc_hip_encode_pack_stereo_frames_async",
    "flacenc_hip_encode_pack_frames_async",
    "flacenc_hip_comm_unique_id",
    "flacenc_hip_comm_create",
    "flacenc_hip_comm_destroy",
    "flacenc_hip_comm_info",
    "flacenc_hip_allgather_async",
    "flacenc_hip_allgather_records_async",
    "flacenc_hip_synchronize",
    "flacenc_sigen_fill_frames",
    "flacenc_sigen_fill_frames_strided",
)


class QlpcConfig(C.Structure):
    """flacenc_hip_qlpc_config == the path's fields of config::Qlpc / config::Prc."""

    _fields_ = [
        ("lpc_order", C.c_uint32),
        ("quant_precision", C.c_uint32),
        ("window_type", C.c_uint32),
        ("tukey_alpha", C.c_float),
        ("max_rice_parameter", C.c_uint32),
        ("flags", C.c_uint32),
        ("use_direct_mse", C.c_uint32),
        ("mae_optimization_steps", C.c_uint32),
    ]


# flacenc_hip_subframe_params (352 bytes)
PARAMS_DTYPE = np.dtype(
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
assert PARAMS_DTYPE.itemsize == 352


class FrameConfig(C.Structure):
    """flacenc_hip_frame_config: the config::Encoder fields that steer encode_frame."""

    _fields_ = [
        ("qlpc", QlpcConfig),
        ("use_constant", C.c_uint32),
        ("use_fixed", C.c_uint32),
        ("use_lpc", C.c_uint32),
        ("use_leftside", C.c_uint32),
        ("use_rightside", C.c_uint32),
        ("use_midside", C.c_uint32),
        ("fixed_max_order", C.c_uint32),
        ("fixed_order_sel", C.c_uint32),
        ("fixed_partitions", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


LAYOUT_SUBFRAMES = 0
LAYOUT_STEREO_FRAMES = 1
ORDERSEL_BITCOUNT = 0
ORDERSEL_APPROXENT = 1
KIND_CONSTANT, KIND_VERBATIM, KIND_FIXED, KIND_LPC = 0, 1, 2, 3


def make_frame_config(qlpc: QlpcConfig | None = None, use_constant=True, use_fixed=False, use_lpc=True,
                      use_leftside=True, use_rightside=True, use_midside=True, fixed_max_order=4,
                      fixed_order_sel=ORDERSEL_APPROXENT, fixed_partitions=16) -> FrameConfig:
    """config::SubFrameCoding / config::Fixed / StereoCoding (src/config.rs:167-183, 236-244, 137-144).
    The reference's default has use_fixed = True; here it is opt-in so that the QLPC-only analysis
    the north-star metric is quoted on stays the default of the bench."""
    return FrameConfig(qlpc or make_config(), int(use_constant), int(use_fixed), int(use_lpc),
                       int(use_leftside), int(use_rightside), int(use_midside), int(fixed_max_order),
                       int(fixed_order_sel), int(fixed_partitions), 0)


# flacenc_hip_channel_result (368 bytes): one channel of an Independent(n) frame
CHANNEL_RESULT_DTYPE = np.dtype(
    [("kind", np.uint8), ("analysis_status", np.uint8), ("pad", np.uint8, (2,)), ("dc_offset", np.int32), ("bits", np.uint64),
     ("params", PARAMS_DTYPE)], align=False)
assert CHANNEL_RESULT_DTYPE.itemsize == 368

# flacenc_hip_stereo_frame_result (752 bytes)
FRAME_RESULT_DTYPE = np.dtype(
    [
        ("channel_assignment", np.uint8),
        ("kind", np.uint8, (2,)),
        ("role", np.uint8, (2,)),
        ("analysis_status", np.uint8),
        ("pad", np.uint8, (2,)),
        ("dc_offset", np.int32, (2,)),
        ("bits", np.uint64, (4,)),
        ("lpc", PARAMS_DTYPE, (2,)),
    ],
    align=True,
)
assert FRAME_RESULT_DTYPE.itemsize == 752


class FlacencHipError(RuntimeError):
    def __init__(self, code, message=""):
        self.code = code
        super().__init__(f"flacenc_hip error {_ERR_NAMES.get(code, code)}: {message}")


_libs = {}


def load() -> C.CDLL:
    """Load libflacenc_hip.so (the product library); raises if it has not been built (no fallback)."""
    return _load_path(LIB_PATH)


def load_hooks() -> C.CDLL:
    """Load libflacenc_hip_hooks.so: the product's objects + flacenc_hip_debug_* (tests and tools only)."""
    return _load_path(HOOKS_LIB_PATH)


def _load_path(LIB_PATH: str) -> C.CDLL:
    if LIB_PATH in _libs:
        return _libs[LIB_PATH]
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
            " or `make -C flacenc_rs_amd/csrc`")
    # One HIP runtime per process: PyTorch ships its own libamdhip64 and the library links the
    # system one under the same SONAME, so whichever is loaded first serves both.  PyTorch cannot
    # work on the system copy ("No HIP GPUs are available"), the library is fine on PyTorch's --
    # so in a process that will use both (tests, bench) PyTorch has to come first.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    vp, i32p, u8p, f64p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
    L.flacenc_hip_abi_version.restype = C.c_int
    # (the config structs are passed by value inside others: a library of another ABI revision reads them shifted)
    if L.flacenc_hip_abi_version() != ABI_VERSION:
        raise RuntimeError("%s has ABI %d, this binding is written for %d" % (LIB_PATH, L.flacenc_hip_abi_version(), ABI_VERSION))
    L.flacenc_hip_device_count.restype = C.c_int
    L.flacenc_hip_create.argtypes = [C.POINTER(vp), C.c_int]
    L.flacenc_hip_create.restype = C.c_int
    L.flacenc_hip_destroy.argtypes = [vp]
    L.flacenc_hip_destroy.restype = None
    L.flacenc_hip_last_error.argtypes = [vp]
    L.flacenc_hip_last_error.restype = C.c_char_p
    L.flacenc_hip_verify_config.argtypes = [C.POINTER(QlpcConfig)]
    L.flacenc_hip_verify_config.restype = C.c_int
    L.flacenc_hip_window_weights.argtypes = [C.POINTER(QlpcConfig), C.c_uint32, vp]
    L.flacenc_hip_window_weights.restype = C.c_int
    L.flacenc_hip_synchronize.argtypes = [vp]
    L.flacenc_hip_synchronize.restype = C.c_int
    # test / profiling hooks (csrc/flacenc_hip_debug.h): present in builds with -DFLACENC_HIP_DEBUG_HOOKS only
    for name in DEBUG_SYMBOLS:
        if hasattr(L, name):
            getattr(L, name).argtypes = [vp, vp]
            getattr(L, name).restype = C.c_int
    if hasattr(L, "flacenc_hip_debug_set_adaptive_order"):
        L.flacenc_hip_debug_set_adaptive_order.argtypes = [vp, C.c_int]
        L.flacenc_hip_debug_adaptive_state.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    batch_args = [vp, C.POINTER(QlpcConfig), i32p, C.c_size_t, C.c_uint32, C.c_size_t, u8p, vp, i32p,
                  C.c_size_t, f64p, f64p]
    L.flacenc_hip_qlpc_batch.argtypes = batch_args + [C.c_int]
    L.flacenc_hip_qlpc_batch.restype = C.c_int
    L.flacenc_hip_qlpc_batch_async.argtypes = batch_args + [vp]
    L.flacenc_hip_qlpc_batch_async.restype = C.c_int
    stereo_args = [vp, C.POINTER(QlpcConfig), i32p, C.c_size_t, C.c_uint32, C.c_size_t, C.c_uint32,
                   vp, i32p, C.c_size_t]
    L.flacenc_hip_stereo_qlpc_batch.argtypes = stereo_args + [C.c_int]
    L.flacenc_hip_stereo_qlpc_batch.restype = C.c_int
    L.flacenc_hip_stereo_qlpc_batch_async.argtypes = stereo_args + [vp]
    L.flacenc_hip_stereo_qlpc_batch_async.restype = C.c_int
    fixed_args = [vp, C.POINTER(FrameConfig), i32p, C.c_size_t, C.c_uint32, C.c_size_t, vp, C.c_uint32,
                  C.c_int, vp, i32p, C.c_size_t, vp]
    L.flacenc_hip_fixed_lpc_batch.argtypes = fixed_args + [C.c_int]
    L.flacenc_hip_fixed_lpc_batch.restype = C.c_int
    L.flacenc_hip_fixed_lpc_batch_async.argtypes = fixed_args + [vp]
    L.flacenc_hip_fixed_lpc_batch_async.restype = C.c_int
    pack_args = [vp, i32p, C.c_size_t, C.c_uint32, C.c_size_t, vp, i32p, C.c_size_t, C.c_uint32, C.c_uint32,
                 C.c_uint32, C.c_uint32, vp, C.c_size_t, vp]
    L.flacenc_hip_pack_stereo_frames.argtypes = pack_args + [C.c_int]
    L.flacenc_hip_pack_stereo_frames.restype = C.c_int
    L.flacenc_hip_pack_stereo_frames_async.argtypes = pack_args + [vp]
    L.flacenc_hip_pack_stereo_frames_async.restype = C.c_int
    L.flacenc_hip_stereo_frame_lengths_async.argtypes = [vp, vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                                          C.c_uint32, C.c_uint32, vp, vp]
    L.flacenc_hip_stereo_frame_lengths_async.restype = C.c_int
    L.flacenc_hip_place_frames_async.argtypes = [vp, vp, vp, vp, C.c_size_t, vp, vp, vp]
    L.flacenc_hip_place_frames_async.restype = C.c_int
    L.flacenc_hip_frame_wire_bytes.argtypes = [C.c_uint32]
    L.flacenc_hip_frame_wire_bytes.restype = C.c_size_t
    L.flacenc_hip_stereo_frame_wire_async.argtypes = [vp, vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                                       C.c_uint32, C.c_uint32, vp, C.c_size_t, vp, vp]
    L.flacenc_hip_stereo_frame_wire_async.restype = C.c_int
    L.flacenc_hip_stream_offsets_async.argtypes = [vp, vp, C.c_size_t, C.c_uint32, C.c_uint64, vp, vp, vp, vp]
    L.flacenc_hip_stream_offsets_async.restype = C.c_int
    L.flacenc_hip_encode_pcm_stereo.argtypes = [vp, C.POINTER(FrameConfig), vp, C.c_uint64, C.c_uint32, C.c_uint32,
                                                C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_size_t, vp,
                                                C.POINTER(C.c_uint64)]
    L.flacenc_hip_encode_pcm_stereo.restype = C.c_int
    L.flacenc_hip_encode_pcm.argtypes = [vp, C.POINTER(FrameConfig), vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_size_t, vp,
                                         C.POINTER(C.c_uint64)]
    L.flacenc_hip_encode_pcm.restype = C.c_int
    L.flacenc_hip_host_alloc.argtypes = [C.c_size_t]
    L.flacenc_hip_host_alloc.restype = C.c_void_p
    L.flacenc_hip_host_free.argtypes = [C.c_void_p]
    L.flacenc_hip_host_free.restype = None
    L.flacenc_hip_set_host_threads.argtypes = [C.c_void_p, C.c_int]
    L.flacenc_hip_set_host_threads.restype = C.c_int
    L.flacenc_hip_fill_le_bytes.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_size_t, C.c_uint32, i32p,
                                            C.c_size_t, C.c_int]
    L.flacenc_hip_fill_le_bytes.restype = C.c_int
    L.flacenc_hip_fill_le_bytes_async.argtypes = [vp, vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_size_t, C.c_uint32,
                                                  vp, C.c_size_t, vp]
    L.flacenc_hip_fill_le_bytes_async.restype = C.c_int
    L.flacenc_hip_encode_pack_stereo_frames_async.argtypes = [vp, C.POINTER(FrameConfig), vp, C.c_size_t, C.c_uint32,
                                                               C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                                               C.c_uint32, vp, vp, C.c_size_t, vp, vp]
    L.flacenc_hip_encode_pack_stereo_frames_async.restype = C.c_int
    L.flacenc_hip_encode_pack_frames_async.argtypes = [vp, C.POINTER(FrameConfig), vp, C.c_size_t, C.c_uint32,
                                                        C.c_uint32, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                                        C.c_uint32, vp, vp, C.c_size_t, vp, vp]
    L.flacenc_hip_encode_pack_frames_async.restype = C.c_int
    L.flacenc_hip_stereo_frame_bytes_bound.argtypes = [C.c_uint32, C.c_uint32]
    L.flacenc_hip_stereo_frame_bytes_bound.restype = C.c_size_t
    L.flacenc_hip_encode_frames.argtypes = [vp, C.POINTER(FrameConfig), i32p, C.c_size_t, C.c_uint32, C.c_uint32,
                                            C.c_size_t, C.c_uint32, vp, i32p, C.c_size_t, C.c_int]
    L.flacenc_hip_encode_frames.restype = C.c_int
    L.flacenc_hip_encode_frames_async.argtypes = L.flacenc_hip_encode_frames.argtypes[:-1] + [vp]
    L.flacenc_hip_encode_frames_async.restype = C.c_int
    L.flacenc_hip_frame_bytes_bound.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    L.flacenc_hip_frame_bytes_bound.restype = C.c_size_t
    L.flacenc_hip_pack_frames.argtypes = [vp, i32p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_size_t, vp, i32p,
                                          C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_size_t,
                                          vp, C.c_int]
    L.flacenc_hip_pack_frames.restype = C.c_int
    L.flacenc_hip_pack_frames_async.argtypes = L.flacenc_hip_pack_frames.argtypes[:-1] + [vp]
    L.flacenc_hip_pack_frames_async.restype = C.c_int
    frame_args = [vp, C.POINTER(FrameConfig), i32p, C.c_size_t, C.c_uint32, C.c_size_t, C.c_uint32,
                  vp, i32p, C.c_size_t]
    L.flacenc_hip_encode_stereo_frames.argtypes = frame_args + [C.c_int]
    L.flacenc_hip_encode_stereo_frames.restype = C.c_int
    L.flacenc_hip_encode_stereo_frames_async.argtypes = frame_args + [vp]
    L.flacenc_hip_encode_stereo_frames_async.restype = C.c_int
    L.flacenc_hip_comm_unique_id.argtypes = [vp]
    L.flacenc_hip_comm_unique_id.restype = C.c_int
    L.flacenc_hip_comm_create.argtypes = [vp, vp, C.c_int, C.c_int]
    L.flacenc_hip_comm_create.restype = C.c_int
    L.flacenc_hip_comm_destroy.argtypes = [vp]
    L.flacenc_hip_comm_destroy.restype = C.c_int
    L.flacenc_hip_comm_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.flacenc_hip_comm_info.restype = C.c_int
    L.flacenc_hip_allgather_async.argtypes = [vp, vp, vp, C.c_size_t, vp]
    L.flacenc_hip_allgather_async.restype = C.c_int
    L.flacenc_hip_allgather_records_async.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, vp, vp]
    L.flacenc_hip_allgather_records_async.restype = C.c_int
    L.flacenc_sigen_fill_frames.argtypes = [vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_size_t,
                                            C.c_uint32, C.c_float, C.c_float, C.c_float,
                                            C.c_uint64, C.c_uint64, C.c_int]
    L.flacenc_sigen_fill_frames.restype = C.c_int
    L.flacenc_sigen_fill_frames_strided.argtypes = [vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_size_t,
                                                    C.c_uint32, C.c_float, C.c_float, C.c_float,
                                                    C.c_uint64, C.c_uint64, C.c_uint64, C.c_int]
    L.flacenc_sigen_fill_frames_strided.restype = C.c_int
    _libs[LIB_PATH] = L
    return L


FLAG_FINEST_RICE_ORDER = 2
FLAG_GENERIC_KERNEL = 4
FLAG_FUSED_PACK = 8
FLAG_TWO_STAGE_PACK = 16
FLAG_REFERENCE_SUM_ORDER = 32
FLAG_NIGHTLY_SUM_ORDER = 64
FLAG_CANONICAL_SUM_ORDER = 128  # the kernels' own order without the order certificate
FLAG_INTEGER_PARITY_ONLY = 256  # with FLAG_REFERENCE_SUM_ORDER: certified shapes keep their own order (integers only)


def make_config(lpc_order=10, quant_precision=15, window=("tukey", 0.4), max_rice_parameter=30,
                flags=0, rice_finest_only=False, use_direct_mse=False, mae_optimization_steps=0) -> QlpcConfig:
    """Defaults of config::Qlpc / config::Prc (src/constant.rs:109-115, src/config.rs:216-221)."""
    if window == "rectangle" or window[0] == "rectangle":
        wt, alpha = WINDOW_RECTANGLE, 0.0
    else:
        wt, alpha = WINDOW_TUKEY, float(window[1])
    if lpc_order > 24:
        flags |= FLAG_ALLOW_ORDER_32
    if rice_finest_only:  # build extension: BASELINE config 2's "fixed Rice partition order"
        flags |= FLAG_FINEST_RICE_ORDER
    return QlpcConfig(lpc_order, quant_precision, wt, alpha, max_rice_parameter, flags, 1 if use_direct_mse else 0,
                      int(mae_optimization_steps))


def verify_config(cfg: QlpcConfig) -> int:
    return int(load().flacenc_hip_verify_config(C.byref(cfg)))


def window_weights(cfg: QlpcConfig, block_size: int) -> np.ndarray:
    out = np.empty(block_size, np.float32)
    rc = load().flacenc_hip_window_weights(C.byref(cfg), block_size, out.ctypes.data)
    if rc != OK:
        raise FlacencHipError(rc)
    return out


def sigen_frames(n_frames: int, channels: int, block_size: int, bits_per_sample: int,
                 sine_period: float, sine_amplitude: float, noise_amplitude: float, seed: int,
                 first_frame: int = 0, nthreads: int | None = None, frame_step: int = 1) -> np.ndarray:
    """flacenc_sigen_fill_frames[_strided] -> int32 [n_frames, channels, block_size] (FrameBuf layout).

    Local frame f is stream frame first_frame + f*frame_step."""
    out = np.empty((n_frames, channels, block_size), np.int32)
    if nthreads is None:
        nthreads = min(16, os.cpu_count() or 1)
    rc = load().flacenc_sigen_fill_frames_strided(out.ctypes.data, n_frames, channels, block_size,
                                                  block_size, bits_per_sample, sine_period,
                                                  sine_amplitude, noise_amplitude, seed, first_frame,
                                                  frame_step, nthreads)
    if rc != 0:
        raise FlacencHipError(rc, "flacenc_sigen_fill_frames")
    return out


def pinned_array(nbytes: int) -> np.ndarray:
    """uint8 array in page-locked host memory (flacenc_hip_host_alloc).  The allocation lives exactly as long
    as the array and its views: the ctypes buffer the array is built on is their common base object, and a
    finalizer on that buffer hands the memory back (flacenc_hip_host_free)."""
    import weakref
    lib = load()
    p = lib.flacenc_hip_host_alloc(nbytes)
    if not p:
        raise MemoryError("flacenc_hip_host_alloc")
    buf = (C.c_uint8 * nbytes).from_address(p)
    weakref.finalize(buf, lib.flacenc_hip_host_free, p)
    arr = np.frombuffer(buf, dtype=np.uint8)
    arr.flags.writeable = True
    return arr


class Handle:
    """RAII wrapper of flacenc_hip_handle (one per host thread / GPU)."""

    def __init__(self, device_id: int = 0, hooks: bool = False):
        """hooks=True: a handle of libflacenc_hip_hooks.so, the only one the debug_* methods below work on."""
        self._lib = load_hooks() if hooks else load()
        self._h = C.c_void_p()
        rc = self._lib.flacenc_hip_create(C.byref(self._h), device_id)
        if rc != OK:
            raise FlacencHipError(rc, "flacenc_hip_create failed (is a GPU visible?)")

    def close(self):
        if self._h:
            self._lib.flacenc_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != OK:
            raise FlacencHipError(rc, self._lib.flacenc_hip_last_error(self._h).decode())

    def _hook(self, name):
        if not hasattr(self._lib, name):
            raise RuntimeError(name + " is not in the product library: make the handle with Handle(dev, hooks=True)")
        return getattr(self._lib, name)

    def debug_set_fixed_keys(self, device_ptr: int):
        self._check(self._hook("flacenc_hip_debug_set_fixed_keys")(self._h, device_ptr or None))

    def debug_set_stamps(self, device_ptr: int):
        self._check(self._hook("flacenc_hip_debug_set_stamps")(self._h, device_ptr or None))

    def debug_set_cert_stats(self, device_ptr: int):
        """3 x uint32 on the device: subframes certified launches analysed, certificates that needed the rows of T^-1,
        subframes recomputed from the reference's chains (flacenc_hip_debug.h)."""
        self._check(self._hook("flacenc_hip_debug_set_cert_stats")(self._h, device_ptr or None))

    def debug_set_adaptive_order(self, on: bool):
        """Launches of the certified shapes that return integers only switch to the two-pass form on hard material
        (flacenc_hip_debug.h); off pins the fused kernel's certificate."""
        self._check(self._hook("flacenc_hip_debug_set_adaptive_order")(self._h, 1 if on else 0))

    def debug_adaptive_state(self):
        """(span, left) of the adaptive order mode: span 0 = the material last seen was easy."""
        span, left = C.c_int(0), C.c_int(0)
        self._check(self._hook("flacenc_hip_debug_adaptive_state")(self._h, C.byref(span), C.byref(left)))
        return span.value, left.value

    def synchronize(self):
        self._check(self._lib.flacenc_hip_synchronize(self._h))

    def set_host_threads(self, threads: int):
        """Host threads sharing the staging copies of encode_pcm* for pageable buffers (default 4)."""
        self._check(self._lib.flacenc_hip_set_host_threads(self._h, threads))

    # -- host-memory path (what the reference's FFI would hand over) ----------
    def qlpc_batch(self, samples, bps, cfg: QlpcConfig, want_fp: bool = False):
        """estimated_qlpc over a [n_subframes, block_size] int32 host array.

        Returns (params record array, residual [n_subframes, block_size], R, a).
        """
        x = np.ascontiguousarray(samples, np.int32)
        ns, n = x.shape
        params = np.zeros(ns, PARAMS_DTYPE)
        residual = np.zeros((ns, n), np.int32)
        bps_a = np.ascontiguousarray(np.broadcast_to(np.asarray(bps, np.uint8), (ns,)))
        R = np.zeros((ns, 33), np.float64) if want_fp else None
        A = np.zeros((ns, 32), np.float64) if want_fp else None
        rc = self._lib.flacenc_hip_qlpc_batch(
            self._h, C.byref(cfg), x.ctypes.data, ns, n, n, bps_a.ctypes.data, params.ctypes.data,
            residual.ctypes.data, n, R.ctypes.data if want_fp else None,
            A.ctypes.data if want_fp else None, MEM_HOST)
        self._check(rc)
        return params, residual, R, A

    def stereo_qlpc_batch(self, frames, bits_per_sample: int, cfg: QlpcConfig):
        """The four estimated_qlpc calls of encode_frame per 2-channel frame (L, R, M, S).

        `frames` is int32 [n_frames, 2, block_size]; returns (params [n_frames, 4],
        residual [n_frames, 4, block_size]).
        """
        x = np.ascontiguousarray(frames, np.int32)
        nf, ch, n = x.shape
        assert ch == 2
        params = np.zeros((nf, 4), PARAMS_DTYPE)
        residual = np.zeros((nf, 4, n), np.int32)
        rc = self._lib.flacenc_hip_stereo_qlpc_batch(
            self._h, C.byref(cfg), x.ctypes.data, nf, n, n, bits_per_sample, params.ctypes.data,
            residual.ctypes.data, n, MEM_HOST)
        self._check(rc)
        return params, residual

    def stereo_qlpc_batch_device(self, cfg: QlpcConfig, frames_ptr: int, n_frames: int,
                                 block_size: int, stride: int, bits_per_sample: int, params_ptr: int,
                                 residual_ptr: int, residual_stride: int, stream: int | None = None):
        rc = self._lib.flacenc_hip_stereo_qlpc_batch_async(
            self._h, C.byref(cfg), frames_ptr, n_frames, block_size, stride, bits_per_sample,
            params_ptr, residual_ptr, residual_stride, stream or None)
        self._check(rc)

    def fixed_lpc_batch(self, samples, bps, cfg: FrameConfig, stereo: bool = False):
        """fixed_lpc (src/coding.rs:298-331) for a batch of any block size.

        plain:  `samples` int32 [n_subframes, n], `bps` array or int -> params [n_subframes], residual
                [n_subframes, n], selector keys [n_subframes];
        stereo: `samples` int32 [n_frames, 2, n], `bps` int -> the same for L, R, M, S: [n_frames, 4, ...]."""
        x = np.ascontiguousarray(samples, np.int32)
        if stereo:
            nf, ch, n = x.shape
            assert ch == 2
            n_units, n_sub, shape = nf, nf * 4, (nf, 4)
            bps_arr, bps_uni = None, int(bps)
        else:
            n_units, n = x.shape
            n_sub, shape = n_units, (n_units,)
            if np.ndim(bps) == 0:
                bps_arr, bps_uni = None, int(bps)
            else:
                bps_arr, bps_uni = np.ascontiguousarray(bps, np.uint8), 16
        params = np.zeros(shape, PARAMS_DTYPE)
        residual = np.zeros(shape + (n,), np.int32)
        keys = np.zeros(shape, np.uint64)
        rc = self._lib.flacenc_hip_fixed_lpc_batch(
            self._h, C.byref(cfg), x.ctypes.data, n_units, n, n,
            bps_arr.ctypes.data if bps_arr is not None else None, bps_uni,
            LAYOUT_STEREO_FRAMES if stereo else LAYOUT_SUBFRAMES, params.ctypes.data, residual.ctypes.data, n,
            keys.ctypes.data, MEM_HOST)
        self._check(rc)
        return params, residual, keys

    def pack_stereo_frames(self, frames, results, residual, bits_per_sample: int, sample_rate: int,
                           first_frame_number: int = 0, frame_number_step: int = 1):
        """Frame::write (src/component/bitrepr.rs:289-319) for the frames encode_stereo_frames decided:
        -> list of `bytes`, one FLAC frame each."""
        x = np.ascontiguousarray(frames, np.int32)
        nf, ch, n = x.shape
        res = np.ascontiguousarray(results)
        rs = np.ascontiguousarray(residual, np.int32)
        stride = int(self._lib.flacenc_hip_stereo_frame_bytes_bound(n, bits_per_sample))
        out = np.zeros((nf, stride), np.uint8)
        lens = np.zeros(nf, np.uint32)
        rc = self._lib.flacenc_hip_pack_stereo_frames(
            self._h, x.ctypes.data, nf, n, n, res.ctypes.data, rs.ctypes.data, n, bits_per_sample, sample_rate,
            first_frame_number, frame_number_step, out.ctypes.data, stride, lens.ctypes.data, MEM_HOST)
        self._check(rc)
        return [bytes(out[f, :lens[f]]) for f in range(nf)]

    def pack_stereo_frames_device(self, frames_ptr: int, n_frames: int, block_size: int, stride: int,
                                  results_ptr: int, residual_ptr: int, residual_stride: int, bits_per_sample: int,
                                  sample_rate: int, first_frame_number: int, frame_number_step: int, out_ptr: int,
                                  out_stride: int, out_len_ptr: int, stream: int | None = None):
        rc = self._lib.flacenc_hip_pack_stereo_frames_async(
            self._h, frames_ptr, n_frames, block_size, stride, results_ptr, residual_ptr, residual_stride,
            bits_per_sample, sample_rate, first_frame_number, frame_number_step, out_ptr, out_stride, out_len_ptr,
            stream or None)
        self._check(rc)

    def stereo_frame_lengths_device(self, results_ptr: int, n_frames: int, block_size: int, bits_per_sample: int,
                                    sample_rate: int, first_frame_number: int, frame_number_step: int,
                                    out_len_ptr: int, stream: int | None = None):
        rc = self._lib.flacenc_hip_stereo_frame_lengths_async(
            self._h, results_ptr, n_frames, block_size, bits_per_sample, sample_rate, first_frame_number,
            frame_number_step, out_len_ptr, stream or None)
        self._check(rc)

    def frame_wire_bytes(self, block_size: int) -> int:
        return int(self._lib.flacenc_hip_frame_wire_bytes(block_size))

    def stereo_frame_wire_device(self, results_ptr: int, n_frames: int, block_size: int, bits_per_sample: int,
                                 sample_rate: int, first_frame_number: int, frame_number_step: int, wire_ptr: int,
                                 wire_stride: int, out_len_ptr: int | None, stream: int | None = None):
        """Decision records -> wire records (+ the frames' byte lengths when out_len_ptr is given), one kernel."""
        rc = self._lib.flacenc_hip_stereo_frame_wire_async(
            self._h, results_ptr, n_frames, block_size, bits_per_sample, sample_rate, first_frame_number,
            frame_number_step, wire_ptr, wire_stride, out_len_ptr or None, stream or None)
        self._check(rc)

    def stream_offsets_device(self, gathered_lengths_ptr: int, n_frames_total: int, world: int, header_bytes: int,
                              lengths_stream_ptr: int | None, offsets_ptr: int, total_ptr: int,
                              stream: int | None = None):
        """All-gathered lengths (rank-major, uint32) -> stream-order lengths (uint32, optional), offsets and total
        (uint64), one kernel."""
        rc = self._lib.flacenc_hip_stream_offsets_async(self._h, gathered_lengths_ptr, n_frames_total, world,
                                                        header_bytes, lengths_stream_ptr or None, offsets_ptr,
                                                        total_ptr, stream or None)
        self._check(rc)

    def place_frames_device(self, src_ptr: int, src_offsets_ptr: int, lengths_ptr: int, n_frames: int, dst_ptr: int,
                            dst_offsets_ptr: int, stream: int | None = None):
        """ParSink's reordering on the GPU: frame i = lengths[i] bytes, src + src_offsets[i] -> dst + dst_offsets[i]
        (uint64 offsets, uint32 lengths, all device pointers)."""
        rc = self._lib.flacenc_hip_place_frames_async(self._h, src_ptr, src_offsets_ptr, lengths_ptr, n_frames,
                                                      dst_ptr, dst_offsets_ptr, stream or None)
        self._check(rc)

    def encode_pcm(self, pcm: np.ndarray, channels: int, cfg: FrameConfig, bytes_per_sample: int, bits_per_sample: int,
                   block_size: int, sample_rate: int, first_frame_number: int = 0, frame_number_step: int = 1):
        """flacenc_hip_encode_pcm: interleaved LE PCM of 1..8 channels -> (frame bytes, lengths)."""
        assert pcm.dtype == np.uint8 and pcm.flags["C_CONTIGUOUS"]
        total = pcm.size // (channels * bytes_per_sample)
        n_frames = (total + block_size - 1) // block_size
        bound = int(self._lib.flacenc_hip_frame_bytes_bound(channels, block_size, bits_per_sample))
        out = np.empty(n_frames * (bound + 16), np.uint8)
        lens = np.zeros(n_frames, np.uint32)
        written = C.c_uint64(0)
        rc = self._lib.flacenc_hip_encode_pcm(self._h, C.byref(cfg), pcm.ctypes.data, total, channels, bytes_per_sample,
                                              bits_per_sample, block_size, sample_rate, first_frame_number,
                                              frame_number_step, out.ctypes.data, out.size, lens.ctypes.data,
                                              C.byref(written))
        self._check(rc)
        return out[: written.value], lens

    def encode_pcm_stereo(self, pcm: np.ndarray, cfg: FrameConfig, bytes_per_sample: int, bits_per_sample: int,
                          block_size: int, sample_rate: int, out: np.ndarray | None = None,
                          first_frame_number: int = 0, frame_number_step: int = 1):
        """Packed interleaved LE stereo PCM (uint8 array) -> (frame bytes uint8 [total], lengths uint32 [n_frames]),
        the streaming host path (flacenc_hip_encode_pcm_stereo).  `pcm` / `out` may be pinned_array()s."""
        assert pcm.dtype == np.uint8 and pcm.flags["C_CONTIGUOUS"]
        total = pcm.size // (2 * bytes_per_sample)
        n_frames = (total + block_size - 1) // block_size
        if out is None:
            out = np.empty(n_frames * (self.frame_bytes_bound(block_size, bits_per_sample) + 16), np.uint8)
        lens = np.zeros(n_frames, np.uint32)
        written = C.c_uint64(0)
        rc = self._lib.flacenc_hip_encode_pcm_stereo(self._h, C.byref(cfg), pcm.ctypes.data, total, bytes_per_sample,
                                                     bits_per_sample, block_size, sample_rate, first_frame_number,
                                                     frame_number_step, out.ctypes.data, out.size, lens.ctypes.data,
                                                     C.byref(written))
        self._check(rc)
        return out[: written.value], lens

    def fill_le_bytes(self, data: bytes, channels: int, bytes_per_sample: int, block_size: int):
        """FrameBuf::fill_le_bytes for a whole stream: packed interleaved PCM -> int32 [n_frames, channels, n]."""
        b = np.frombuffer(bytes(data), np.uint8).copy()
        total = len(b) // (channels * bytes_per_sample)
        nf = (total + block_size - 1) // block_size
        out = np.full((nf, channels, block_size), -1, np.int32)
        rc = self._lib.flacenc_hip_fill_le_bytes(self._h, b.ctypes.data, total, channels, bytes_per_sample, nf,
                                                 block_size, out.ctypes.data, block_size, MEM_HOST)
        self._check(rc)
        return out

    def encode_pack_stereo_frames_device(self, cfg: FrameConfig, frames_ptr: int, n_frames: int, block_size: int,
                                         stride: int, bits_per_sample: int, sample_rate: int,
                                         first_frame_number: int, frame_number_step: int, results_ptr: int,
                                         out_ptr: int, out_stride: int, out_len_ptr: int, stream: int | None = None):
        """PCM frames -> FLAC frame bytes on the device (one fused kernel for 4096-sample blocks)."""
        rc = self._lib.flacenc_hip_encode_pack_stereo_frames_async(
            self._h, C.byref(cfg), frames_ptr, n_frames, block_size, stride, bits_per_sample, sample_rate,
            first_frame_number, frame_number_step, results_ptr, out_ptr, out_stride, out_len_ptr, stream or None)
        self._check(rc)

    def encode_pack_frames_device(self, cfg: FrameConfig, frames_ptr: int, n_frames: int, channels: int,
                                  block_size: int, stride: int, bits_per_sample: int, sample_rate: int,
                                  first_frame_number: int, frame_number_step: int, results_ptr: int, out_ptr: int,
                                  out_stride: int, out_len_ptr: int, stream: int | None = None):
        rc = self._lib.flacenc_hip_encode_pack_frames_async(
            self._h, C.byref(cfg), frames_ptr, n_frames, channels, block_size, stride, bits_per_sample, sample_rate,
            first_frame_number, frame_number_step, results_ptr, out_ptr, out_stride, out_len_ptr, stream or None)
        self._check(rc)

    def frame_bytes_bound(self, block_size: int, bits_per_sample: int) -> int:
        return int(self._lib.flacenc_hip_stereo_frame_bytes_bound(block_size, bits_per_sample))

    # -- the ordered gather's collective (RCCL communicator owned by the handle) ----------
    @staticmethod
    def comm_unique_id() -> bytes:
        """ncclGetUniqueId: 128 bytes rank 0 hands to the other ranks."""
        buf = (C.c_uint8 * COMM_ID_BYTES)()
        rc = load().flacenc_hip_comm_unique_id(C.cast(buf, C.c_void_p))
        if rc != OK:
            raise FlacencHipError(rc, "flacenc_hip_comm_unique_id")
        return bytes(buf)

    def comm_create(self, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == COMM_ID_BYTES
        buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
        self._check(self._lib.flacenc_hip_comm_create(self._h, C.cast(buf, C.c_void_p), rank, world))

    def comm_destroy(self):
        self._check(self._lib.flacenc_hip_comm_destroy(self._h))

    def comm_info(self):
        r, w = C.c_int(0), C.c_int(0)
        self._check(self._lib.flacenc_hip_comm_info(self._h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def allgather_device(self, send_ptr: int, recv_ptr: int, bytes_per_rank: int, stream: int | None = None):
        self._check(self._lib.flacenc_hip_allgather_async(self._h, send_ptr, recv_ptr, bytes_per_rank, stream or None))

    def allgather_records_device(self, local_ptr: int, n_local: int, n_total: int, record_bytes: int, gathered_ptr: int,
                                 stream: int | None = None):
        """The ordered gather's exchange: rank-major, short ranks zero-padded (all_gather_rank_major's layout)."""
        self._check(self._lib.flacenc_hip_allgather_records_async(self._h, local_ptr or None, n_local, n_total, record_bytes,
                                                                  gathered_ptr, stream or None))

    def frame_bytes_bound_channels(self, channels: int, block_size: int, bits_per_sample: int) -> int:
        return int(self._lib.flacenc_hip_frame_bytes_bound(channels, block_size, bits_per_sample))

    def encode_frames_device(self, cfg: FrameConfig, frames_ptr: int, n_frames: int, channels: int, block_size: int,
                             stride: int, bits_per_sample: int, results_ptr: int, residual_ptr: int,
                             residual_stride: int, stream: int | None = None):
        """flacenc_hip_encode_frames_async: Independent(channels) frames, device pointers."""
        rc = self._lib.flacenc_hip_encode_frames_async(
            self._h, C.byref(cfg), frames_ptr, n_frames, channels, block_size, stride, bits_per_sample, results_ptr,
            residual_ptr, residual_stride, stream or None)
        self._check(rc)

    def pack_frames_device(self, frames_ptr: int, n_frames: int, channels: int, block_size: int, stride: int,
                           results_ptr: int, residual_ptr: int, residual_stride: int, bits_per_sample: int,
                           sample_rate: int, first_frame_number: int, frame_number_step: int, out_ptr: int,
                           out_stride: int, out_len_ptr: int, stream: int | None = None):
        """flacenc_hip_pack_frames_async: Frame::write for the frames encode_frames_device decided."""
        rc = self._lib.flacenc_hip_pack_frames_async(
            self._h, frames_ptr, n_frames, channels, block_size, stride, results_ptr, residual_ptr, residual_stride,
            bits_per_sample, sample_rate, first_frame_number, frame_number_step, out_ptr, out_stride, out_len_ptr,
            stream or None)
        self._check(rc)

    def encode_frames(self, frames, bits_per_sample: int, cfg: FrameConfig):
        """encode_frame for Independent(channels) frames: `frames` int32 [n_frames, channels, n] ->
        (results CHANNEL_RESULT_DTYPE [n_frames, channels], residual [n_frames, channels, n])."""
        x = np.ascontiguousarray(frames, np.int32)
        nf, ch, n = x.shape
        results = np.zeros((nf, ch), CHANNEL_RESULT_DTYPE)
        residual = np.zeros((nf, ch, n), np.int32)
        rc = self._lib.flacenc_hip_encode_frames(self._h, C.byref(cfg), x.ctypes.data, nf, ch, n, n, bits_per_sample,
                                                 results.ctypes.data, residual.ctypes.data, n, MEM_HOST)
        self._check(rc)
        return results, residual

    def pack_frames(self, frames, results, residual, bits_per_sample: int, sample_rate: int,
                    first_frame_number: int = 0, frame_number_step: int = 1):
        """Frame::write for the frames encode_frames decided -> list of `bytes`."""
        x = np.ascontiguousarray(frames, np.int32)
        nf, ch, n = x.shape
        res = np.ascontiguousarray(results)
        rs = np.ascontiguousarray(residual, np.int32)
        stride = int(self._lib.flacenc_hip_frame_bytes_bound(ch, n, bits_per_sample))
        out = np.zeros((nf, stride), np.uint8)
        lens = np.zeros(nf, np.uint32)
        rc = self._lib.flacenc_hip_pack_frames(self._h, x.ctypes.data, nf, ch, n, n, res.ctypes.data, rs.ctypes.data, n,
                                               bits_per_sample, sample_rate, first_frame_number, frame_number_step,
                                               out.ctypes.data, stride, lens.ctypes.data, MEM_HOST)
        self._check(rc)
        return [bytes(out[f, :lens[f]]) for f in range(nf)]

    def encode_stereo_frames(self, frames, bits_per_sample: int, cfg: FrameConfig):
        """encode_frame with the decision on the GPU: `frames` int32 [n_frames, 2, block_size] ->
        (results FRAME_RESULT_DTYPE [n_frames], residual [n_frames, 2, block_size])."""
        x = np.ascontiguousarray(frames, np.int32)
        nf, ch, n = x.shape
        assert ch == 2
        results = np.zeros(nf, FRAME_RESULT_DTYPE)
        residual = np.zeros((nf, 2, n), np.int32)
        rc = self._lib.flacenc_hip_encode_stereo_frames(
            self._h, C.byref(cfg), x.ctypes.data, nf, n, n, bits_per_sample, results.ctypes.data,
            residual.ctypes.data, n, MEM_HOST)
        self._check(rc)
        return results, residual

    def encode_stereo_frames_device(self, cfg: FrameConfig, frames_ptr: int, n_frames: int,
                                    block_size: int, stride: int, bits_per_sample: int,
                                    results_ptr: int, residual_ptr: int, residual_stride: int,
                                    stream: int | None = None):
        rc = self._lib.flacenc_hip_encode_stereo_frames_async(
            self._h, C.byref(cfg), frames_ptr, n_frames, block_size, stride, bits_per_sample,
            results_ptr, residual_ptr, residual_stride, stream or None)
        self._check(rc)

    # -- device-memory path (raw pointers; torch tensors' data_ptr()) ---------
    def qlpc_batch_device(self, cfg: QlpcConfig, samples_ptr: int, n_subframes: int, block_size: int,
                          stride: int, bps_ptr: int, params_ptr: int, residual_ptr: int,
                          residual_stride: int, autocorr_ptr: int = 0, lpc_ptr: int = 0,
                          stream: int | None = None, sync: bool = False):
        if stream is None and sync:
            rc = self._lib.flacenc_hip_qlpc_batch(
                self._h, C.byref(cfg), samples_ptr, n_subframes, block_size, stride, bps_ptr or None,
                params_ptr, residual_ptr, residual_stride, autocorr_ptr or None, lpc_ptr or None,
                MEM_DEVICE)
        else:
            rc = self._lib.flacenc_hip_qlpc_batch_async(
                self._h, C.byref(cfg), samples_ptr, n_subframes, block_size, stride, bps_ptr or None,
                params_ptr, residual_ptr, residual_stride, autocorr_ptr or None, lpc_ptr or None,
                stream or None)
        self._check(rc)
