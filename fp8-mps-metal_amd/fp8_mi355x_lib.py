"""
ctypes binding of libfp8mi.so - the C-ABI HIP library declared in
include/fp8mi.h.

This is the only place the product touches native code.  It is the MI355X
counterpart of the reference's two kernel loaders: the lazy
`torch.mps.compile_shader` singleton (fp8_mps_native.py:14-38) and the pybind11
module `fp8_metal` (fp8_bridge.cpp:361-371).  Differences by design:

  * the library is built ahead of time for gfx950 (`make -C fp8-mps-metal_amd`
    or `__graft_entry__.build()`); nothing is compiled at run time;
  * entry points take raw device pointers and a HIP stream, so tensors are
    never staged through the CPU (fp8_bridge.cpp:180-258 does exactly that);
  * there is NO CPU fallback: if the library is missing or a launch fails this
    module raises, loudly.
"""

from __future__ import annotations

import ctypes
import os
import threading

# torch first: libfp8mi.so links libamdhip64, and PyTorch-ROCm ships its own copy of that
# runtime.  Whichever is loaded first serves the whole process; loading ours first and
# torch's afterwards leaves two HIP runtimes in one process ("no ROCm-capable device").
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# FP8MI_LIB_PATH lets a diagnostic build (e.g. libfp8mi_stamp.so) stand in; default is the product library
LIB_PATH = os.environ.get("FP8MI_LIB_PATH") or os.path.join(_HERE, "libfp8mi.so")

# enums of include/fp8mi.h
F32, F16, BF16 = 0, 1, 2
SCALE_TENSOR, SCALE_ROW = 0, 1
NAN_ZERO, NAN_PROPAGATE = 0, 1
ENC_REFERENCE, ENC_RNE = 0, 1
KERNEL_AUTO, KERNEL_GEMV, KERNEL_GEMM_128, KERNEL_GENERIC, KERNEL_GEMM_256, KERNEL_GEMM_128x64, KERNEL_SKINNY = range(7)
KERNEL_GEMM_64x128 = 14
KERNEL_GEMV_FP32 = 18
KERNEL_GEMV_MX = 19
KERNEL_GEMM_256W = 20
KERNEL_GEMM_256x128W = 21
KERNEL_GEMM_64x64 = 22
KERNEL_GEMM_32x64 = 23
KERNEL_GEMM_32x32 = 24
KERNEL_GEMM_128D = 25
WS_COUNTER_BYTES = 4096
EPILOGUE_TRANSPOSED = 0x100  # OR into bias_dtype (include/fp8mi.h)

_vp, _i64, _int = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int


class DeviceInfo(ctypes.Structure):
    _fields_ = [
        ("compute_units", _int), ("clock_khz", _int), ("memory_clock_khz", _int),
        ("memory_bus_bits", _int), ("l2_bytes", _int), ("lds_bytes_per_cu", _int),
        ("wavefront_size", _int), ("total_memory", _i64),
        ("arch", ctypes.c_char * 64), ("name", ctypes.c_char * 128),
    ]


# symbol -> (restype, argtypes); every function include/fp8mi.h declares
SIGNATURES = {
    "fp8mi_scaled_mm": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64,
                               _int, _int, _int, _int, _int, _vp]),
    "fp8mi_scaled_mm_ex": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64,
                                  _int, _int, _int, _int, _int, _int, _vp]),
    "fp8mi_scaled_mm_ws": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64,
                                  _int, _int, _int, _int, _int, _int, _int, _vp, _i64, _vp]),
    "fp8mi_scaled_mm_workspace_bytes": (_i64, []),
    "fp8mi_workspace_reset": (_int, [_vp, _i64, _vp]),
    "fp8mi_choose_kernel": (_int, [_i64, _i64, _i64, _i64, _i64, _i64, _int, _int, _int]),
    "fp8mi_predict_kernel_us": (ctypes.c_double, [_int, _i64, _i64, _i64, _i64, _i64, _i64, _int, _int, _int, _int]),
    "fp8mi_dequant": (_int, [_vp, _vp, _vp, _i64, _int, _vp]),
    "fp8mi_dequant_f16": (_int, [_vp, _vp, _vp, _i64, _int, _vp]),   # SURVEY 8(b)'s name, same entry point
    "fp8mi_encode": (_int, [_vp, _int, _vp, _vp, _i64, _int, _vp]),
    "fp8mi_amax": (_int, [_vp, _int, _vp, _i64, _vp]),
    "fp8mi_quantize": (_int, [_vp, _int, _vp, _vp, _i64, _int, _vp]),
    "fp8mi_device_info": (_int, [_int, ctypes.POINTER(DeviceInfo)]),
    "fp8mi_profile_begin": (_int, [_int]),
    "fp8mi_profile_end": (_int, [ctypes.POINTER(ctypes.c_float), _int]),
    "fp8mi_version": (_int, []),
    "fp8mi_last_error": (ctypes.c_char_p, []),
}

_lib = None
_lock = threading.Lock()


class Fp8miError(RuntimeError):
    """A libfp8mi call returned non-zero (argument error < 0, hipError_t > 0)."""


def load():
    """Load libfp8mi.so once (thread-safe) and return the ctypes handle.

    Raises RuntimeError if the library has not been built - the product has no
    other way to compute anything, by design.
    """
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"fp8mi: {LIB_PATH} not found. Build it with `make -C {_HERE}` "
                "(hipcc --offload-arch=gfx950) or `python -c 'import __graft_entry__ as g; g.build()'`. "
                "There is no CPU fallback."
            )
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().fp8mi_last_error()
        raise Fp8miError(f"{what} failed (code {rc}): {msg.decode(errors='replace') if msg else ''}")


class kernel_timer:
    """Context manager around fp8mi_profile_begin/_end: collects the pure
    device duration (ms) of every kernel the library launches from this
    thread inside the block, in launch order, into `.ms`."""

    def __init__(self, max_launches: int):
        self.n = int(max_launches)
        self.ms = []

    def __enter__(self):
        check(load().fp8mi_profile_begin(self.n), "fp8mi_profile_begin")
        return self

    def __exit__(self, *exc):
        buf = (ctypes.c_float * self.n)()
        got = load().fp8mi_profile_end(buf, self.n)
        if got < 0:
            check(got, "fp8mi_profile_end")
        self.ms = [float(buf[i]) for i in range(min(got, self.n))]
        return False


def device_info(device: int = 0) -> dict:
    info = DeviceInfo()
    check(load().fp8mi_device_info(device, ctypes.byref(info)), "fp8mi_device_info")
    return {k: (getattr(info, k).decode() if isinstance(getattr(info, k), bytes) else getattr(info, k))
            for k, _ in DeviceInfo._fields_}
