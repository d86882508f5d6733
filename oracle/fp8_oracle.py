"""
CPU oracle (numpy) for the FP8 e4m3fn scaled-matmul + cast path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product (`fp8-mps-metal_amd/`) may
import this module; only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` use it, and only as the checker.

It restates, on the CPU, the algorithm of the reference's four Metal kernels
and their Python glue.  Every function cites the reference lines it follows
(paths are relative to the reference repo root, audiohacking/fp8-mps-metal):

  decode   fp8_matmul.metal:19-40   == test_fp8_correctness.py:22-50
  encode   fp8_matmul.metal:44-92   == test_fp8_correctness.py:53-106
  matmul   fp8_matmul.metal:99-147  (vecmat :155-210 is the M == 1 case)
  dequant  fp8_matmul.metal:215-223 + fp8_mps_native.py:98-124
  quantize fp8_matmul.metal:228-236 + fp8_mps_native.py:127-190
  epilogue fp8_mps_patch.py:94-104

Parity pinning: `tests/golden/` holds vectors produced by importing the
reference's own executable spec (test_fp8_correctness.py) in the build
container (script: tests/golden/make_golden.py); tests/test_oracle.py checks
every function below against them, against the reference's known-answer
tests, and against torch-CPU float8_e4m3fn casts as an independent opinion.
"""

from __future__ import annotations

import numpy as np

FP8_MAX = 448.0

# ----------------------------------------------------------------------------
# decode
# ----------------------------------------------------------------------------


def decode_scalar(bits: int) -> float:
    """One byte -> float, reference semantics (fp8_matmul.metal:19-40).

    NaN patterns 0x7F / 0xFF decode to +0.0 (metal:21); the sign is applied
    last (metal:39) so 0x80 decodes to -0.0.
    """
    if (bits & 0x7F) == 0x7F:
        return 0.0
    sign = (bits >> 7) & 1
    e = (bits >> 3) & 0xF
    m = bits & 0x7
    if e == 0:
        v = (m / 8.0) * 2.0 ** -6
    else:
        v = (1.0 + m / 8.0) * 2.0 ** (e - 7)
    return -v if sign else v


def decode_lut(nan_to_zero: bool = True) -> np.ndarray:
    """float32[256] decode table.  nan_to_zero=False gives OCP/torch semantics
    (0x7F/0xFF -> NaN), which is what the gfx950 hardware converts implement."""
    lut = np.array([decode_scalar(b) for b in range(256)], dtype=np.float32)
    if not nan_to_zero:
        lut[0x7F] = np.nan
        lut[0xFF] = np.nan
    return lut


_LUT = decode_lut()
_LUT64 = _LUT.astype(np.float64)


def decode(u8: np.ndarray) -> np.ndarray:
    """uint8 array -> float32 array (reference semantics)."""
    return _LUT[np.asarray(u8, dtype=np.uint8)]


# ----------------------------------------------------------------------------
# encode (reference semantics)
# ----------------------------------------------------------------------------


def encode_scalar(val: float) -> int:
    """Scalar restatement in float arithmetic, following the reference's
    executable spec line by line in meaning (test_fp8_correctness.py:53-106;
    shader twin fp8_matmul.metal:44-92).  `val` must be an exactly
    representable float32 (callers pass float(np.float32(x))); NaN is outside
    the reference's domain (its spec raises) and is mapped to 0x7F here.
    """
    import math

    if val != val:
        return 0x7F
    sign = 0
    if val < 0.0:  # -0.0 keeps sign 0 (metal:46)
        sign = 1
        val = -val
    if val >= FP8_MAX:  # saturate, also +-inf (metal:53-55)
        return (sign << 7) | 0x7E
    if val < 1.0 / 512.0:  # flush below the smallest subnormal (metal:58-60)
        return sign << 7
    if val < 1.0 / 64.0:  # subnormal, RNE, clamp to 7 (metal:64-70)
        mant = min(int(round(val * 512.0)), 7)
        return (sign << 7) | mant
    e = int(math.floor(math.log2(val)))  # metal:73
    e = max(-6, min(8, e))
    mant = min(int(round((val / 2.0 ** e - 1.0) * 8.0)), 7)  # no carry (metal:79-81)
    eb = max(1, min(15, e + 7))
    if eb == 15 and mant == 7:  # never emit the NaN pattern (metal:87-89)
        mant = 6
    return (sign << 7) | (eb << 3) | mant


def encode(x: np.ndarray) -> np.ndarray:
    """float32 array -> uint8, reference semantics, integer-only on the fp32
    bit pattern (vectorised; identical results to encode_scalar).

    Differences from an OCP / torch-CPU cast, all inherited from the
    reference (fp8_matmul.metal:44-92):
      * a mantissa that rounds up to 8 is clamped to 7 (no carry),
      * (2^-10, 2^-9) flushes to zero instead of rounding to 0x01,
      * anything >= 448 (and +-inf) saturates to 0x7E / 0xFE, never NaN,
      * -0.0 -> 0x00 (the sign is taken from `val < 0`).
    """
    x = np.ascontiguousarray(x, dtype=np.float32)
    bits = x.view(np.uint32)
    a = bits & np.uint32(0x7FFFFFFF)
    neg = ((bits >> np.uint32(31)) == 1) & (a != 0)
    sign = np.where(neg, np.uint32(0x80), np.uint32(0))

    e = (a >> np.uint32(23)).astype(np.int64)  # biased fp32 exponent
    man = (a & np.uint32(0x7FFFFF)).astype(np.int64)

    # normal fp8 range: 3 mantissa bits = top 3 of the 23, RNE on the low 20
    q = man >> 20
    rem = man & 0xFFFFF
    q = q + ((rem > 0x80000) | ((rem == 0x80000) & ((q & 1) == 1)))
    q = np.minimum(q, 7)
    eb = e - 127 + 7
    q = np.where((eb == 15) & (q == 7), 6, q)
    normal = (eb << 3) | q

    # subnormal fp8 range [2^-9, 2^-6): mant = RNE(val * 512), clamp 7
    full = man | 0x800000
    s = np.clip(141 - e, 1, 40)  # val*512 = full * 2^(e-141)
    qs = full >> s
    rs = full & ((np.int64(1) << s) - 1)
    half = np.int64(1) << (s - 1)
    qs = qs + ((rs > half) | ((rs == half) & ((qs & 1) == 1)))
    sub = np.minimum(qs, 7)

    out = np.where(a < np.uint32(0x3C800000), sub, normal)  # < 2^-6
    out = np.where(a < np.uint32(0x3B000000), 0, out)  # < 2^-9
    out = np.where(a >= np.uint32(0x43E00000), 0x7E, out)  # >= 448
    out = out.astype(np.uint32) | sign
    out = np.where(a > np.uint32(0x7F800000), np.uint32(0x7F), out)  # NaN in
    return out.astype(np.uint8)


def encode_torch_rne(x: np.ndarray) -> np.ndarray:
    """float32 -> uint8 with OCP e4m3fn round-to-nearest-even semantics, i.e.
    what torch-CPU `.to(torch.float8_e4m3fn)` produces (non-default mode of
    the build; the second opinion of test_mps_vs_cpu.py:283-357)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    bits = x.view(np.uint32)
    a = (bits & np.uint32(0x7FFFFFFF)).astype(np.int64)
    sign = ((bits >> np.uint32(24)) & np.uint32(0x80)).astype(np.int64)
    e = a >> 23
    man = a & 0x7FFFFF
    # normal: re-bias then RNE on 20 bits with carry into the exponent
    v = ((e - 120) << 23) | man  # exponent field re-biased to e4m3 (7)
    lsb = (v >> 20) & 1
    n = (v + 0x7FFFF + lsb) >> 20
    # subnormal target (fp32 exp < 121): value * 2^9 rounded RNE
    full = man | 0x800000
    s = np.clip(141 - e, 1, 62)
    qs = full >> s
    rs = full & ((np.int64(1) << s) - 1)
    half = np.int64(1) << (s - 1)
    qs = qs + ((rs > half) | ((rs == half) & ((qs & 1) == 1)))
    out = np.where(e < 121, qs, n)
    out = np.where(e < 100, 0, out)  # far below half the smallest subnormal
    out = np.where(out > 0x7E, 0x7F, out)  # overflow -> NaN (no inf in e4m3fn)
    out = np.where(a > 0x7F800000, 0x7F, out)
    return (out | sign).astype(np.uint8)


# ----------------------------------------------------------------------------
# scaled matmul
# ----------------------------------------------------------------------------


def _bcast_scale(s, n, axis_len_name):
    s = np.asarray(s, dtype=np.float32).reshape(-1)
    if s.size == 1:
        return np.full(n, s[0], dtype=np.float32)
    if s.size != n:
        raise ValueError(f"scale has {s.size} elements, expected 1 or {n} ({axis_len_name})")
    return s


def scaled_mm(A_u8, B_nk_u8, scale_a, scale_b, accumulate="f32"):
    """C[m,n] = (sum_k dec(A[m,k]) * dec(B[n,k])) * sa[m or 0] * sb[n or 0].

    Follows fp8_matmul.metal:116-146 (and :173-209 for M == 1): products of
    decoded bytes summed over K, the two scales applied once after the sum,
    as (sum * sa) * sb in float32.  A is (M,K), B is (N,K), both row-major
    uint8 (metal:100-102).  `accumulate="f64"` sums in float64 (products of
    two e4m3 values are exact, so this is the exact dot product up to 2^-53)
    and is what the GPU result is measured against; "f32" is the reference's
    own arithmetic ("CPU dequant + fp32 matmul", BASELINE.json configs[0]).

    Scale broadcasting: a 1-element scale is per-tensor, an M- (resp. N-)
    element scale is per-row.  (The reference indexes a 1-element scale with
    the row when only the other one is per-row - fp8_mps_native.py:73,
    metal:144-146 - which reads out of bounds; the build broadcasts instead.)
    """
    A_u8 = np.asarray(A_u8, dtype=np.uint8)
    B_nk_u8 = np.asarray(B_nk_u8, dtype=np.uint8)
    M, K = A_u8.shape
    N, K2 = B_nk_u8.shape
    if K != K2:
        raise ValueError("K mismatch")
    sa = _bcast_scale(scale_a, M, "M")
    sb = _bcast_scale(scale_b, N, "N")
    if accumulate == "f64":
        acc = (_LUT64[A_u8] @ _LUT64[B_nk_u8].T)
        return (acc * sa.astype(np.float64)[:, None]) * sb.astype(np.float64)[None, :]
    acc = _LUT[A_u8] @ _LUT[B_nk_u8].T
    return ((acc * sa[:, None]) * sb[None, :]).astype(np.float32)


def abs_dot_bound(A_u8, B_nk_u8, scale_a, scale_b):
    """sum_k |dec(A)| |dec(B)| * |sa| * |sb| - the magnitude against which
    float32 accumulation error is measured in the parity tests."""
    A_u8 = np.asarray(A_u8, dtype=np.uint8)
    B_nk_u8 = np.asarray(B_nk_u8, dtype=np.uint8)
    M = A_u8.shape[0]
    N = B_nk_u8.shape[0]
    sa = np.abs(_bcast_scale(scale_a, M, "M")).astype(np.float64)
    sb = np.abs(_bcast_scale(scale_b, N, "N")).astype(np.float64)
    acc = np.abs(_LUT64[A_u8]) @ np.abs(_LUT64[B_nk_u8]).T
    return acc * sa[:, None] * sb[None, :]


def epilogue(C_f32, bias=None, scale_result=None):
    """bias add then result scale, in that order, on the float32 product
    (fp8_mps_patch.py:94-100).  The out_dtype cast (:103-104) is left to the
    caller (torch / numpy casting, round-to-nearest-even)."""
    out = np.asarray(C_f32, dtype=np.float32)
    if bias is not None:
        out = out + np.asarray(bias, dtype=np.float32)[None, :]
    if scale_result is not None:
        out = out * np.float32(np.asarray(scale_result, dtype=np.float32).reshape(-1)[0])
    return out.astype(np.float32)


# ----------------------------------------------------------------------------
# casts
# ----------------------------------------------------------------------------


def dequantize_f16(u8, scale=1.0):
    """half(dec(b)) * half(scale), computed in float16
    (fp8_matmul.metal:215-223, fp8_mps_native.py:114-122)."""
    h = decode(u8).astype(np.float16)  # exact: every e4m3 value fits fp16
    s = np.float16(np.float32(scale))
    with np.errstate(over="ignore"):      # a product beyond 65504 IS inf in fp16, as in the reference's torch multiply
        return (h * s).astype(np.float16)


def quantize(x):
    """Amax-scaled quantisation (fp8_mps_native.py:158-190):
    amax = max|x| (read back as a Python float), scale = 448/amax (double),
    bytes = enc(float32(x) * float32(scale)), returns (bytes, float32(1/scale))."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    amax = float(np.max(np.abs(x))) if x.size else 0.0
    scale = FP8_MAX / amax if amax > 0 else 1.0
    scaled = (x * np.float32(scale)).astype(np.float32)
    return encode(scaled), np.float32(1.0 / scale)


def rel_rmse(x, ref):
    x = np.asarray(x, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    d = np.sqrt(np.mean((x - ref) ** 2))
    r = np.sqrt(np.mean(ref ** 2))
    return d / r if r > 0 else d
