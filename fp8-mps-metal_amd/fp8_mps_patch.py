"""
Drop-in monkey-patch surface for FP8 e4m3fn on MI355X (PyTorch-ROCm).

Same module name and same public surface as the reference's fp8_mps_patch.py,
so a ComfyUI / diffusers call site that did

    import fp8_mps_patch
    fp8_mps_patch.install()

keeps working unchanged; what runs underneath is the hand-written gfx950 HIP
library (libfp8mi.so) instead of the Metal shader:

    install() / uninstall() / is_installed()       fp8_mps_patch.py:443-497
    _metal_scaled_mm   -> torch._scaled_mm         fp8_mps_patch.py:53-106
    _metal_tensor_to   -> torch.Tensor.to          fp8_mps_patch.py:109-226
    _metal_tensor_copy -> torch.Tensor.copy_       fp8_mps_patch.py:229-302
    _original_scaled_mm / _original_tensor_to / _original_tensor_copy
                                                    fp8_mps_patch.py:38-41

(The `_metal_*` names are kept because the reference's tests look them up by
name - test_mps_limits_patch.py:135-143.)

Interception rule: a call is routed to the HIP kernels only when the tensor
lives on a HIP device (`device.type == "cuda"` under PyTorch-ROCm; the
reference tests `"mps"`) AND float8_e4m3fn (or raw uint8 bytes for _scaled_mm)
is involved; every other call reaches the saved original with its arguments
untouched.  There is no CPU fallback anywhere in this module.

Deliberate differences from the reference (each a reference defect, see
DESIGN.md "Boundary"): float8_e5m2 is never intercepted (the reference decodes
e5m2 bytes as e4m3, fp8_mps_patch.py:65); fp8 -> other-fp8 casts go to torch
(the reference reinterprets the bytes, :204-206); a dtype given together with
a device move of an fp8 tensor is honoured (the reference drops it, :160-174);
bias / scale_result / out_dtype are fused into the kernel epilogue instead of
three extra passes (:94-104); no environment variable is set (:451-452) and no
VAE tiling is installed (:362-440) - both work around MPS limits that do not
exist on ROCm.
"""

from __future__ import annotations

import threading

import torch

_original_scaled_mm = None
_original_tensor_to = None
_original_tensor_copy = None
_installed = False

_lock = threading.RLock()
_DEV = "cuda"  # PyTorch-ROCm's device type for HIP GPUs
_E4M3 = getattr(torch, "float8_e4m3fn", None)
_E5M2 = getattr(torch, "float8_e5m2", None)
_DEQUANT_DIRECT = (torch.float16, torch.float32, torch.bfloat16)


def _is_fp8_dtype(dtype):
    """True for either FP8 dtype (fp8_mps_patch.py:44-50)."""
    return dtype is not None and (dtype == _E4M3 or dtype == _E5M2)


def _is_e4m3(dtype):
    return dtype is not None and dtype == _E4M3


_native_mod = None


def _native():
    global _native_mod
    if _native_mod is None:
        import fp8_mi355x_native  # lazy, like the reference's `import fp8_mps_native` (:74)
        _native_mod = fp8_mi355x_native
    return _native_mod


def _is_dev(device) -> bool:
    """Does `device` (str / int / torch.device / None) name a HIP device?"""
    if device is None:
        return False
    try:
        return torch.device(device).type == _DEV
    except (RuntimeError, TypeError):
        return False


# ---------------------------------------------------------------------------
# torch._scaled_mm
# ---------------------------------------------------------------------------

def _metal_scaled_mm(input, other, *args, out_dtype=None, scale_a=None, scale_b=None, bias=None,
                     scale_result=None, use_fast_accum=False):
    """Replacement for torch._scaled_mm (fp8_mps_patch.py:53-106).

    input (M,K) and other (K,N, column-major as torch requires) hold e4m3fn
    values (float8_e4m3fn or raw uint8 bytes).  Returns
    ((input @ other) * scale_a * scale_b + bias) * scale_result as `out_dtype`
    (float32 when None, as the reference).  Scales may also be passed
    positionally, as torch >= 2.5 does.
    """
    if args:  # torch order: scale_a, scale_b, bias, scale_result, out_dtype, use_fast_accum
        names = ("scale_a", "scale_b", "bias", "scale_result", "out_dtype", "use_fast_accum")
        if len(args) > len(names):
            raise TypeError("_scaled_mm() takes at most 8 positional arguments")
        pos = dict(zip(names, args))
        scale_a = pos.get("scale_a", scale_a)
        scale_b = pos.get("scale_b", scale_b)
        bias = pos.get("bias", bias)
        scale_result = pos.get("scale_result", scale_result)
        out_dtype = pos.get("out_dtype", out_dtype)
        use_fast_accum = pos.get("use_fast_accum", use_fast_accum)

    ok = (torch.uint8, _E4M3)
    take = (isinstance(input, torch.Tensor) and isinstance(other, torch.Tensor)
            and input.device.type == _DEV and input.dtype in ok and other.dtype in ok)
    if not take:
        return _original_scaled_mm(input, other, out_dtype=out_dtype, scale_a=scale_a, scale_b=scale_b,
                                   bias=bias, scale_result=scale_result, use_fast_accum=use_fast_accum)

    native = _native()
    dev = input.device
    if scale_a is None:
        scale_a = _ones(dev)
    if scale_b is None:
        scale_b = _ones(dev)
    # The kernels' fused epilogue writes float32 / float16 / bfloat16.  The reference ends in `result.to(out_dtype)` for ANY dtype
    # (fp8_mps_patch.py:103-104) - a float8_e4m3fn result then goes through its patched `.to`, i.e. the encode kernel: same here, as a
    # second launch behind the float32 product.
    final = None
    if out_dtype is not None and out_dtype not in (torch.float32, torch.float16, torch.bfloat16):
        final, out_dtype = out_dtype, torch.float32
    # other is (K,N); the kernels want the (N,K) row-major operand.  For the column-major `other` torch mandates the
    # storage IS that operand: pointers and strides go down as they are - no uint8 views, no .t() (each a tensor
    # construction of ~1 us on a path whose kernels take 5-15 us)
    r = native.scaled_mm_colmajor(input, other, scale_a, scale_b, bias=bias, scale_result=scale_result, out_dtype=out_dtype)
    if r is None:
        # any other layout (row-major `other`, strided rows): the general entry makes the operands contiguous
        a = input if input.dtype == torch.uint8 else input.view(torch.uint8)
        o = other if other.dtype == torch.uint8 else other.view(torch.uint8)
        r = native.fp8_scaled_mm_auto(a, o.t(), scale_a, scale_b, bias=bias, scale_result=scale_result, out_dtype=out_dtype)
    return r if final is None else _metal_tensor_to(r, final)


_ones_cache = {}


def _ones(dev):
    t = _ones_cache.get(dev)
    if t is None:
        t = torch.ones(1, dtype=torch.float32, device=dev)
        _ones_cache[dev] = t
    return t


# ---------------------------------------------------------------------------
# Tensor.to
# ---------------------------------------------------------------------------

def _parse_to_args(args, kwargs):
    """-> (dtype, device, passthrough_kwargs) for the overloads of Tensor.to:
    to(dtype), to(device), to(device, dtype), to(other), keyword forms, plus
    positional non_blocking / copy flags."""
    dtype = kwargs.get("dtype")
    device = kwargs.get("device")
    extra = {k: v for k, v in kwargs.items() if k not in ("dtype", "device")}
    flags = []
    for a in args:
        if isinstance(a, torch.dtype):
            if dtype is None:
                dtype = a
        elif isinstance(a, torch.Tensor):
            if dtype is None:
                dtype = a.dtype
            if device is None:
                device = a.device
        elif isinstance(a, bool):
            flags.append(a)
        elif isinstance(a, (torch.device, str, int)):
            if device is None:
                device = a
    for name, val in zip(("non_blocking", "copy"), flags):
        extra.setdefault(name, val)
    return dtype, device, extra


def _to_scenario(src_dtype, src_on_dev: bool, dtype, device) -> str:
    """Pure routing decision of _metal_tensor_to (fp8_mps_patch.py:160-226):
      "bytes_to_device"  fp8 tensor elsewhere -> HIP device: move raw bytes
      "encode"           non-fp8 -> float8_e4m3fn on a HIP device: encode kernel
      "same"             e4m3 on device, nothing to change
      "dequant"          e4m3 on device -> float dtype: dequant kernel
      "original"         everything else: saved torch.Tensor.to
    """
    target_on_dev = _is_dev(device) if device is not None else src_on_dev
    src_fp8 = _is_fp8_dtype(src_dtype)
    if src_fp8 and device is not None and target_on_dev and not src_on_dev:
        if dtype is None or dtype == src_dtype or (_is_e4m3(src_dtype) and not _is_fp8_dtype(dtype)):
            return "bytes_to_device"
        return "original"
    if target_on_dev and _is_e4m3(dtype) and not src_fp8:
        return "encode"
    if _is_e4m3(src_dtype) and src_on_dev and (device is None or target_on_dev):
        if dtype is None or dtype == src_dtype:
            return "same"
        if not _is_fp8_dtype(dtype) and dtype.is_floating_point:
            return "dequant"
    return "original"


def _dequant_to(t_fp8, dtype):
    native = _native()
    u8 = t_fp8.view(torch.uint8)
    if dtype in _DEQUANT_DIRECT:
        return native.fp8_dequantize(u8, None, out_dtype=dtype)
    # e.g. float64: every e4m3 value is exact in float32, widen from there
    return _original_tensor_to(native.fp8_dequantize(u8, None, out_dtype=torch.float32), dtype)


def _metal_tensor_to(self, *args, **kwargs):
    """Replacement for Tensor.to (fp8_mps_patch.py:109-226): FP8 conversions
    that touch a HIP device go through the encode / dequant kernels, with the
    reference's value-preserving semantics (no scaling); the rest is torch's."""
    dtype, device, extra = _parse_to_args(args, kwargs)
    scenario = _to_scenario(self.dtype, self.device.type == _DEV, dtype, device)

    if scenario == "original":
        return _original_tensor_to(self, *args, **kwargs)

    if scenario == "bytes_to_device":
        moved = _original_tensor_to(self.view(torch.uint8), device, **extra).view(self.dtype)
        if dtype is not None and dtype != self.dtype:
            return _dequant_to(moved, dtype)
        return moved

    if scenario == "encode":
        # move first (original .to, dtype untouched), then encode on the device
        src = _original_tensor_to(self, device, **extra) if device is not None else self
        return _native().fp8_encode(src).view(dtype)

    if scenario == "same":
        if device is not None:  # e.g. cuda:0 -> cuda:1: raw bytes, torch returns self when nothing changes
            return _original_tensor_to(self.view(torch.uint8), device, **extra).view(self.dtype)
        return self.clone() if extra.get("copy") else self

    # "dequant"
    src = self
    if device is not None:
        src = _original_tensor_to(self.view(torch.uint8), device).view(self.dtype)
    return _dequant_to(src, dtype)


# ---------------------------------------------------------------------------
# Tensor.copy_
# ---------------------------------------------------------------------------

def _copy_scenario(dst_dtype, dst_on_dev: bool, src_dtype) -> str:
    """Routing decision of _metal_tensor_copy (fp8_mps_patch.py:250-302):
      "bytes"    same fp8 dtype -> fp8 destination on a HIP device: byte copy
      "encode"   non-fp8 source -> float8_e4m3fn destination on device
      "original" everything else
    """
    if not dst_on_dev:
        return "original"
    if _is_fp8_dtype(dst_dtype) and src_dtype == dst_dtype:
        return "bytes"
    if _is_e4m3(dst_dtype) and not _is_fp8_dtype(src_dtype):
        return "encode"
    return "original"


def _metal_tensor_copy(self, src, non_blocking=False):
    """Replacement for Tensor.copy_ (fp8_mps_patch.py:229-302).  Returns self."""
    if not isinstance(src, torch.Tensor):
        return _original_tensor_copy(self, src, non_blocking=non_blocking)
    scenario = _copy_scenario(self.dtype, self.device.type == _DEV, src.dtype)
    if scenario == "bytes":
        _original_tensor_copy(self.view(torch.uint8), src.contiguous().view(torch.uint8), non_blocking=non_blocking)
        return self
    if scenario == "encode":
        s = src if src.device == self.device else _original_tensor_to(src, self.device)
        _original_tensor_copy(self.view(torch.uint8), _native().fp8_encode(s), non_blocking=non_blocking)
        return self
    return _original_tensor_copy(self, src, non_blocking=non_blocking)


# ---------------------------------------------------------------------------
# install / uninstall
# ---------------------------------------------------------------------------

def patch_vae_decode_for_mps_limits():
    """Kept for surface compatibility (test_mps_limits_patch.py:135-143).  The
    reference tiles comfy.sd.VAE.decode to stay under MPSGraph's INT_MAX tensor
    limit (fp8_mps_patch.py:362-440); ROCm has no such limit, so this is a no-op."""
    return None


def install():
    """Swap torch._scaled_mm, Tensor.to and Tensor.copy_ for the HIP-backed
    versions (idempotent; fp8_mps_patch.py:443-471)."""
    global _original_scaled_mm, _original_tensor_to, _original_tensor_copy, _installed
    with _lock:
        if _installed:
            return
        if not hasattr(torch, "_scaled_mm"):
            raise RuntimeError("torch._scaled_mm not found — requires PyTorch 2.4+")
        _original_scaled_mm = torch._scaled_mm
        _original_tensor_to = torch.Tensor.to
        _original_tensor_copy = torch.Tensor.copy_
        torch._scaled_mm = _metal_scaled_mm
        torch.Tensor.to = _metal_tensor_to
        torch.Tensor.copy_ = _metal_tensor_copy
        patch_vae_decode_for_mps_limits()
        _installed = True


def uninstall():
    """Restore the exact original objects (fp8_mps_patch.py:474-492)."""
    global _original_scaled_mm, _original_tensor_to, _original_tensor_copy, _installed
    with _lock:
        if not _installed:
            return
        if _original_scaled_mm is not None:
            torch._scaled_mm = _original_scaled_mm
            _original_scaled_mm = None
        if _original_tensor_to is not None:
            torch.Tensor.to = _original_tensor_to
            _original_tensor_to = None
        if _original_tensor_copy is not None:
            torch.Tensor.copy_ = _original_tensor_copy
            _original_tensor_copy = None
        _ones_cache.clear()
        _installed = False


def is_installed():
    """Is the monkey-patch active? (fp8_mps_patch.py:495-497)"""
    return _installed
