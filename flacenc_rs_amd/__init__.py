"""flacenc_rs_amd -- MI355X-native QLPC analysis path of flacenc-rs behind a C ABI.

The product is ``libflacenc_hip.so`` (HIP kernels + ``extern "C"`` entry points
declared in ``include/flacenc_hip.h``).  This package holds its sources
(``csrc/``) and the thin ctypes plumbing the Python test and bench drivers use.
"""
from . import _capi  # noqa: F401
from ._capi import Handle, QlpcConfig, make_config, FlacencHipError, PARAMS_DTYPE  # noqa: F401
