"""
FP8 e4m3fn ops on MI355X: the op layer between the monkey-patch
(fp8_mps_patch.py) and the HIP kernels (libfp8mi.so through fp8_mi355x_lib).

It mirrors the reference's op module fp8_mps_native.py function for function -
same names, same argument meaning, same error behaviour - so that code and
tests written against the reference read the same here:

  fp8_scaled_mm       fp8_mps_native.py:41-95
  fp8_dequantize      fp8_mps_native.py:98-124
  fp8_encode          fp8_mps_native.py:127-155
  fp8_quantize        fp8_mps_native.py:158-190
  fp8_scaled_mm_auto  fp8_mps_native.py:193-210
  fp8_scaled_mm_fast  fp8_mps_native.py:213-267 (kept as an alias: the
                      dequant -> fp16 matmul detour it implemented exists only
                      because Apple GPUs have no FP8 ALU; on gfx950 the MFMA
                      kernel consumes the bytes directly)

What differs, deliberately:
  * the device is "cuda" (PyTorch-ROCm's name for HIP devices), not "mps";
  * kernels are launched on torch's CURRENT stream of the tensor's device and
    never synchronise (the reference's `.item()` in fp8_quantize,
    fp8_mps_native.py:174, is gone: amax, scale and encode all stay on the GPU);
  * bias / scale_result / out_dtype can be passed down and are fused into the
    kernel epilogue (the reference applies them as three extra passes,
    fp8_mps_patch.py:94-104);
  * a per-row scale next to a per-tensor scale is broadcast properly (the
    reference reads out of bounds in that case, fp8_mps_native.py:73 with
    fp8_matmul.metal:144-146);
  * there is no CPU path.  A tensor that is not on a HIP device is moved there
    (as the reference moves to "mps", fp8_mps_native.py:63-66); without a GPU
    that raises.
"""

from __future__ import annotations

import threading

import torch

import fp8_mi355x_lib as _l

DEVICE_TYPE = "cuda"  # PyTorch-ROCm reports HIP devices as "cuda"

_DTYPE_CODE = {torch.float32: _l.F32, torch.float16: _l.F16, torch.bfloat16: _l.BF16}

# process-wide defaults; see include/fp8mi.h for the meaning of the modes
NAN_MODE = _l.NAN_ZERO          # reference decode: NaN bytes are 0.0
ENCODE_MODE = _l.ENC_REFERENCE  # reference encode rules


# Conversions made INSIDE this module never involve an fp8 dtype, so they use the C-level Tensor.to: while the
# monkey-patch is installed `tensor.to(...)` is fp8_mps_patch._metal_tensor_to, a Python function that parses the
# overloads of .to before it can decide that the call is none of its business (~3 us per call on the hot path).
_TO = torch._C.TensorBase.to

_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device):
    """hipStream_t (as int) of torch's current stream on `device`.  The raw accessor skips the construction of a
    torch.cuda.Stream object (1.5 us of the ~4 us this module adds to a call; tools/time_callsite.py)."""
    if _raw_stream is not None:
        return _raw_stream(device.index if device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(device).cuda_stream


class _on_device:
    """`with torch.cuda.device(dev)` only when `dev` is not already current (the context manager costs two device
    switches and ~2 us even when it changes nothing)."""
    __slots__ = ("ctx",)

    def __init__(self, dev):
        self.ctx = None if (dev.index is None or torch.cuda.current_device() == dev.index) else torch.cuda.device(dev)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


# split-K workspaces: one per (device, stream) - launches on one stream are ordered, so they can share it; the
# counters at its head are zeroed once here and left zero by every launch (include/fp8mi.h).  Kept alive for the
# life of the process, so a pointer captured into a HIP graph stays valid.
_workspaces: dict = {}
_workspaces_lock = threading.Lock()


def _workspace(device):
    """The split-K workspace of (device, current stream), or None while that stream is being captured into a HIP graph
    and no workspace exists yet: allocating inside a capture would put it into the graph's private pool, where later
    eager launches on the same stream handle would share it (warm the op up once before capturing; without a workspace
    the call simply does not split K)."""
    return _workspace_on(device, _stream(device))


def _workspace_on(device, stream):
    key = (device.index, stream)
    ws = _workspaces.get(key)
    if ws is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        with _workspaces_lock:
            ws = _workspaces.get(key)
            if ws is None:
                ws = torch.empty(int(_l.load().fp8mi_scaled_mm_workspace_bytes()), dtype=torch.uint8, device=device)
                _l.check(_l.load().fp8mi_workspace_reset(ws.data_ptr(), ws.numel(), stream), "fp8mi_workspace_reset")
                _workspaces[key] = ws
    return ws


def reset_workspaces():
    """Zero the arrival counters of every cached split-K workspace (include/fp8mi.h, fp8mi_workspace_reset): only needed
    after a launch was aborted mid-flight (device fault, process-level error recovery) - every completed launch leaves
    them zero by itself."""
    with _workspaces_lock:
        for (dev_index, stream), ws in _workspaces.items():
            with torch.cuda.device(dev_index):
                # a workspace is ordered on the stream it is keyed by: the reset goes onto THAT stream (not the caller's
                # current one), bracketed by device syncs so that it neither races a launch in flight nor trails the next
                torch.cuda.synchronize(dev_index)
                _l.check(_l.load().fp8mi_workspace_reset(ws.data_ptr(), ws.numel(), stream), "fp8mi_workspace_reset")
                torch.cuda.synchronize(dev_index)


def _to_device(t: torch.Tensor) -> torch.Tensor:
    return t if t.device.type == DEVICE_TYPE else t.to(DEVICE_TYPE)


def _scale_arg(scale, device, rows: int, what: str):
    """-> (float32 contiguous tensor on `device`, mode).  1 element = per-tensor,
    `rows` elements (any shape, e.g. (M,1) / (1,N)) = per-row."""
    s = scale
    # already what the kernel reads (the common case: a float32 device scalar or (M,1) / (1,N) column): no new tensor
    if not (s.dtype is torch.float32 and s.device == device and s.is_contiguous()):
        s = _TO(s, device=device, dtype=torch.float32).reshape(-1).contiguous()
    if s.numel() == 1:
        return s, _l.SCALE_TENSOR
    if s.numel() == rows:
        return s, _l.SCALE_ROW
    raise AssertionError(f"{what} has {s.numel()} elements; expected 1 or {rows}")


def fp8_scaled_mm(A: torch.Tensor, B: torch.Tensor, scale_a: torch.Tensor, scale_b: torch.Tensor,
                  *, bias: torch.Tensor | None = None, scale_result: torch.Tensor | None = None,
                  out_dtype: torch.dtype | None = None, nan_mode: int | None = None,
                  kernel: int = _l.KERNEL_AUTO, split_k: int = 0, out: torch.Tensor | None = None,
                  transposed_epilogue: bool = False) -> torch.Tensor:
    """FP8 scaled matrix multiplication on the GPU.

    A: (M, K) uint8 - e4m3fn bytes, row-major
    B: (N, K) uint8 - e4m3fn bytes, row-major (i.e. pre-transposed); a row
       stride larger than K is accepted without a copy
    scale_a: [1] or [M] float32;  scale_b: [1] or [N] float32
    Returns (M, N) float32 (or `out_dtype`) on the device:
        ((A_dec @ B_dec.T) * scale_a * scale_b + bias) * scale_result
    Same contract as fp8_mps_native.py:41-95 (asserts included); the kernel is
    picked by shape inside the library (GEMV for M == 1, MFMA GEMM otherwise).
    split_k: 0 lets the library slice K when M x N gives too few tiles to fill
    the GPU (small batch, deep K), 1 forbids it, > 1 forces that many slices.
    out: optional (M, N) destination of `out_dtype` with unit column stride (any
    row stride: e.g. a slab of a larger buffer); returned instead of a new tensor.
    transposed_epilogue: the call computes C^T = W . X^T for a caller whose
    activations are `B` here (fp8_sharded_linear): `bias` then has M elements and
    runs along the rows, and the scales are applied in the order of the
    untransposed product (include/fp8mi.h, FP8MI_EPILOGUE_TRANSPOSED).
    """
    assert A.dtype == torch.uint8 and B.dtype == torch.uint8
    assert A.dim() == 2 and B.dim() == 2
    M, K = A.shape
    N = B.shape[0]
    assert B.shape[1] == K

    A = _to_device(A)
    B = _to_device(B)
    dev = A.device
    assert B.device == dev, "A and B must be on the same device"
    # rows must be dense in K; a padded row stride is fine (no copy)
    if not (K == 0 or M == 0 or (A.stride(1) == 1 and A.stride(0) >= K) or (M == 1 and A.stride(1) == 1)):
        A = A.contiguous()
    if not (K == 0 or N == 0 or (B.stride(1) == 1 and B.stride(0) >= K) or (N == 1 and B.stride(1) == 1)):
        B = B.contiguous()
    lda = max(A.stride(0), K) if M > 1 else max(K, 1)
    ldb = max(B.stride(0), K) if N > 1 else max(K, 1)
    # The MFMA / vec-mat kernels read 16-byte pieces: K, lda, ldb multiples of 16 and 16-byte aligned bases.  Anything else the reference
    # accepts (fp8_mps_native.py:55-60 asks for contiguity only: K = 4100, a sliced weight view) would run on the library's generic kernel -
    # one wave per output element, orders of magnitude slower (M=N=4096, K=4100: profiles/r04_unaligned.txt).  Large such problems are
    # copied once per call into aligned buffers whose rows are padded with ZERO bytes up to the next multiple of 16: a zero byte is +0.0
    # in e4m3, so the padded product is the same sum (tests/test_gpu_parity.py::test_unaligned_operands_take_the_padded_mfma_path).
    if kernel == _l.KERNEL_AUTO and K > 0 and M >= 2 and M * N * K >= PAD_MIN_MACS and not (_aligned16(A, M, K, lda) and _aligned16(B, N, K, ldb)):
        A, B, K, lda, ldb = _pad_operands(A, B, M, N, K)
    return _scaled_mm_core(A, B, M, N, K, lda, ldb, dev, scale_a, scale_b, bias, scale_result, out_dtype, nan_mode,
                           kernel, split_k, out, transposed_epilogue)


PAD_MIN_MACS = 1 << 22   # below this many multiply-adds the generic kernel is as fast as two extra copy launches; a single row (M = 1) never pays for
                         # a copy of the whole weight matrix (K=4100 N=4096: generic 22.6 us, padded 33.0: profiles/r04_unaligned.txt)


def _aligned16(t, rows, K, ld):
    return K % 16 == 0 and t.data_ptr() % 16 == 0 and (rows <= 1 or ld % 16 == 0)


def _pad_operands(A, B, M, N, K):
    """-> (A', B', K', lda', ldb'): fresh, contiguous (so 16-byte aligned) copies with rows zero-padded to K' = the next multiple of 16."""
    Kp = (K + 15) // 16 * 16
    if Kp != K:
        pad = torch.nn.functional.pad
        return pad(A, (0, Kp - K)), pad(B, (0, Kp - K)), Kp, Kp, Kp
    return A.contiguous().clone() if A.data_ptr() % 16 else A.contiguous(), B.contiguous().clone() if B.data_ptr() % 16 else B.contiguous(), K, K, K


def _scaled_mm_core(a_keep, b_keep, M, N, K, lda, ldb, dev, scale_a, scale_b, bias, scale_result, out_dtype, nan_mode,
                    kernel, split_k, out, transposed_epilogue):
    """Everything behind the operand checks: scales, output, epilogue arguments, ONE ctypes call.  `a_keep` / `b_keep`
    are tensors of ANY dtype whose storage holds the (M,K) / (N,K) byte rows at data_ptr() with row strides lda / ldb
    (the patch hands over the float8 tensors themselves: no uint8 views, no .t())."""
    sa, sa_mode = _scale_arg(scale_a, dev, M, "scale_a")
    sb, sb_mode = _scale_arg(scale_b, dev, N, "scale_b")

    out_dtype = torch.float32 if out_dtype is None else out_dtype
    out_code = _DTYPE_CODE.get(out_dtype)
    if out_code is None:
        raise AssertionError(f"unsupported out_dtype {out_dtype}")
    if out is not None:
        assert out.shape == (M, N) and out.dtype == out_dtype and out.device == dev, "out must be (M, N) out_dtype on A's device"
        assert N <= 1 or out.stride(1) == 1, "out needs unit column stride"
        C = out
    else:
        C = torch.empty((M, N), dtype=out_dtype, device=dev)
    if M == 0 or N == 0:
        return C
    ldc = max(C.stride(0), N) if M > 1 else max(N, 1)

    bias_ptr, bias_code = None, _l.F32
    if bias is not None:
        if bias.device != dev:
            bias = _TO(bias, device=dev)
        if bias.dtype not in _DTYPE_CODE:
            bias = _TO(bias, torch.float32)
        if not bias.is_contiguous():
            bias = bias.reshape(-1).contiguous()
        nb = M if transposed_epilogue else N
        assert bias.numel() == nb, f"bias has {bias.numel()} elements; expected {nb}"
        bias_ptr, bias_code = bias.data_ptr(), _DTYPE_CODE[bias.dtype]
    if transposed_epilogue:
        bias_code |= _l.EPILOGUE_TRANSPOSED
    sr_ptr = None
    if scale_result is not None:
        if not (scale_result.dtype is torch.float32 and scale_result.device == dev and scale_result.is_contiguous()):
            scale_result = _TO(scale_result, device=dev, dtype=torch.float32).reshape(-1).contiguous()
        assert scale_result.numel() == 1, "scale_result must have one element"
        sr_ptr = scale_result.data_ptr()

    lib = _l.load()
    with _on_device(dev):
        stream = _stream(dev)
        # a workspace only where split-K can apply: more than one row, K deep enough to slice
        ws = _workspace_on(dev, stream) if (split_k != 1 and M > 1 and K >= 1024) else None
        rc = lib.fp8mi_scaled_mm_ws(
            a_keep.data_ptr(), b_keep.data_ptr(), C.data_ptr(), sa.data_ptr(), sb.data_ptr(), bias_ptr, sr_ptr,
            M, N, K, lda, ldb, ldc, sa_mode, sb_mode, out_code, bias_code,
            NAN_MODE if nan_mode is None else nan_mode, kernel, split_k if ws is not None else 1,
            ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, stream)
    if rc:
        _l.check(rc, "fp8mi_scaled_mm")
    return C


def scaled_mm_colmajor(input: torch.Tensor, other: torch.Tensor, scale_a, scale_b, *, bias=None, scale_result=None,
                       out_dtype=None):
    """The call torch._scaled_mm makes, without intermediate tensors: `input` (M,K) row-major and `other` (K,N) in the
    column-major layout torch mandates, both float8_e4m3fn or uint8 ON A HIP DEVICE.  `other`'s storage then already is
    the (N,K) row-major operand the kernels read (fp8_mps_patch.py:77-86 makes it with .t().contiguous()), so the
    pointers and strides are passed as they are.  Returns None when the layout is anything else (the caller falls back
    to fp8_scaled_mm, which copies)."""
    if input.dim() != 2 or other.dim() != 2:
        return None
    M, K = input.shape
    K2, N = other.shape
    if K2 != K or other.device != input.device:
        return None
    sa0, sa1 = input.stride()
    sb0, sb1 = other.stride()
    if not (K == 0 or M == 0 or (sa1 == 1 and sa0 >= K) or (M == 1 and sa1 == 1)):
        return None
    if not (K == 0 or N == 0 or (sb0 == 1 and sb1 >= K) or (N == 1 and sb0 == 1)):
        return None
    lda = max(sa0, K) if M > 1 else max(K, 1)
    ldb = max(sb1, K) if N > 1 else max(K, 1)
    return _scaled_mm_core(input, other, M, N, K, lda, ldb, input.device, scale_a, scale_b, bias, scale_result, out_dtype,
                           None, _l.KERNEL_AUTO, 0, None, False)


def fp8_dequantize(input: torch.Tensor, scale: torch.Tensor | None = None,
                   out_dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """FP8 -> half dequantisation on the GPU (fp8_mps_native.py:98-124).

    input: uint8 tensor (e4m3fn bytes);  scale: scalar tensor or None.
    Returns float16 (or `out_dtype`) of the same shape:
    half(decode(b)) * half(scale), the product formed in float16 as in the
    reference; float32 / bfloat16 outputs convert that half value.
    """
    input = _to_device(input)
    assert input.dtype == torch.uint8
    dev = input.device
    src = input.contiguous()
    out = torch.empty(input.shape, dtype=out_dtype, device=dev)
    count = src.numel()
    if count == 0:
        return out
    s_ptr = None
    if scale is not None:
        scale = _TO(scale, device=dev, dtype=torch.float32).reshape(-1).contiguous()
        assert scale.numel() == 1, "scale must be a scalar"
        s_ptr = scale.data_ptr()
    lib = _l.load()
    with _on_device(dev):
        rc = lib.fp8mi_dequant(src.data_ptr(), out.data_ptr(), s_ptr, count, _DTYPE_CODE[out_dtype], _stream(dev))
    _l.check(rc, "fp8mi_dequant")
    return out


def _encode_source(input: torch.Tensor) -> torch.Tensor:
    inp = _to_device(input)
    if inp.dtype not in _DTYPE_CODE:  # ints, float64 ...: the reference converts to float32 first
        inp = inp.to(torch.float32)
    return inp.contiguous()


def fp8_encode(input: torch.Tensor, encode_mode: int | None = None) -> torch.Tensor:
    """Float -> FP8 bytes without scaling (fp8_mps_native.py:127-155): values
    keep their magnitude, saturating at +-448.  Used by .to(float8_e4m3fn) and
    .copy_().  float16 / bfloat16 sources are widened inside the kernel.
    Returns uint8 of the same shape."""
    inp = _encode_source(input)
    dev = inp.device
    out = torch.empty(inp.shape, dtype=torch.uint8, device=dev)
    count = inp.numel()
    if count == 0:
        return out
    lib = _l.load()
    with _on_device(dev):
        rc = lib.fp8mi_encode(inp.data_ptr(), _DTYPE_CODE[inp.dtype], out.data_ptr(), None, count,
                              ENCODE_MODE if encode_mode is None else encode_mode, _stream(dev))
    _l.check(rc, "fp8mi_encode")
    return out


def fp8_quantize(input: torch.Tensor, encode_mode: int | None = None):
    """Float -> FP8 with automatic amax scaling (fp8_mps_native.py:158-190).

    Returns (uint8 tensor, inverse_scale[1] float32) with
    scale = 448 / max|input|; everything (amax, scale, encode) runs on the
    device, no host read-back."""
    inp = _encode_source(input)
    dev = inp.device
    out = torch.empty(inp.shape, dtype=torch.uint8, device=dev)
    scales = torch.empty(2, dtype=torch.float32, device=dev)
    lib = _l.load()
    with _on_device(dev):
        rc = lib.fp8mi_quantize(inp.data_ptr(), _DTYPE_CODE[inp.dtype], out.data_ptr(), scales.data_ptr(),
                                inp.numel(), ENCODE_MODE if encode_mode is None else encode_mode, _stream(dev))
    _l.check(rc, "fp8mi_quantize")
    return out, scales[1:2]


def fp8_linear(x: torch.Tensor, weight_u8: torch.Tensor, weight_scale: torch.Tensor, bias: torch.Tensor | None = None,
               out_dtype: torch.dtype | None = None) -> torch.Tensor:
    """y = x @ dequant(W).T + bias with dynamic per-tensor activation quantisation - the composition the reference's
    call sites perform around its two entry points (fp8_quantize, fp8_mps_native.py:158-190, then torch._scaled_mm
    through fp8_mps_patch.py:53-106), as one call: amax + scaled encode of x (two launches, no host sync), then the
    scaled matmul with the fused bias / cast epilogue.

    x: (..., K) float32 / float16 / bfloat16;  weight_u8: (N, K) e4m3fn bytes;  weight_scale: [1] or [N] float32.
    Returns (..., N) in `out_dtype` (default: x.dtype, float32 for other inputs)."""
    assert weight_u8.dtype == torch.uint8 and weight_u8.dim() == 2
    K = weight_u8.shape[1]
    assert x.shape[-1] == K, f"x has {x.shape[-1]} features; weight expects {K}"
    x2 = _to_device(x).reshape(-1, K)
    xq, x_inv_scale = fp8_quantize(x2)
    if out_dtype is None:
        out_dtype = x.dtype if x.dtype in _DTYPE_CODE else torch.float32
    y = fp8_scaled_mm(xq, weight_u8, x_inv_scale, weight_scale, bias=bias, out_dtype=out_dtype)
    return y.reshape(*x.shape[:-1], weight_u8.shape[0])


def pad_weight_rows(weight: torch.Tensor, pad_bytes: int = 256) -> torch.Tensor:
    """The same (N, K) fp8 / uint8 weight in a buffer whose ROW STRIDE is K + pad_bytes (a one-time copy at load time).  No counterpart
    in the reference (its kernels take no strides); the C ABI and every Python entry point here take the stride as it is (`ldb`):
    pass the returned view, or its `.t()` to the patched torch._scaled_mm.  Results are bit-identical to the unpadded call.

    Why: the small-batch tile kernels stream W as 8-row x 128-byte pieces, and with some power-of-two-ish row strides those pieces
    crowd onto few memory channels.  Measured on MI355X for M <= 128 (tools/sweep_pad.py, profiles/r03_row_stride.txt), +256 bytes:
    K = 16384, N = 4096: 18.5 -> 14.3 us at M = 16 (19.6 -> 16.0 at M = 64); K = N = 8192: 14.2 -> 12.4 (18.1 -> 16.8);
    K = N = 16384: 55.6 -> 45.9; K = 20480, N = 4096: 21.0 -> 17.0 at M = 64.  It is NOT a rule of K alone: K = 8192 against
    N = 4096 or 14336, K = 4096, 14336, 24576, 28672 and every M >= 512 do not change, and K = 32768 gets 23 % SLOWER with 256
    (unchanged with 512).  Measure the deployment's shapes with tools/sweep_pad.py before padding a model's weights."""
    assert weight.dim() == 2 and weight.element_size() == 1, "an (N, K) matrix of fp8 bytes"
    assert pad_bytes >= 0 and pad_bytes % 16 == 0, "the tile kernels need 16-byte aligned rows"
    if pad_bytes == 0:
        return weight
    N, K = weight.shape
    buf = torch.empty((N, K + pad_bytes), dtype=torch.uint8, device=weight.device)
    view = buf[:, :K]
    view.copy_(weight.view(torch.uint8) if weight.dtype != torch.uint8 else weight)
    return view if weight.dtype == torch.uint8 else view.view(weight.dtype)


def fp8_amax(input: torch.Tensor) -> torch.Tensor:
    """max|input| as a float32[1] device tensor (no host sync)."""
    inp = _encode_source(input)
    dev = inp.device
    out = torch.empty(1, dtype=torch.float32, device=dev)
    lib = _l.load()
    with _on_device(dev):
        rc = lib.fp8mi_amax(inp.data_ptr(), _DTYPE_CODE[inp.dtype], out.data_ptr(), inp.numel(), _stream(dev))
    _l.check(rc, "fp8mi_amax")
    return out


def fp8_scaled_mm_auto(A, B, scale_a, scale_b, **kw):
    """Shape-based strategy choice (fp8_mps_native.py:193-210).  The reference
    switches at M <= 16 between its fused kernel and a dequant + fp16 matmul;
    here the choice (GEMV / MFMA GEMM tile shape / generic) is made inside
    fp8mi_scaled_mm from M, N, K and alignment."""
    return fp8_scaled_mm(A, B, scale_a, scale_b, **kw)


# The reference's "fast" path (fp8_mps_native.py:213-267) dequantises both
# operands to fp16 and calls the native matmul.  On gfx950 that would be slower
# and less accurate than the fp8 MFMA kernel, so the name maps to the same op.
fp8_scaled_mm_fast = fp8_scaled_mm
