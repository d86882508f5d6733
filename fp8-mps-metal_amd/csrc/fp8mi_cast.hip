// Elementwise FP8 e4m3fn casts for gfx950: dequant (fp8 -> f16/f32/bf16),
// encode (f32/f16/bf16 -> fp8), amax, and the device-side scale derivation of
// the amax-scaled quantiser.  All are HBM-bound streaming kernels: 16 bytes of
// fp8 per lane per step, dwordx4 loads and stores, grid-stride over at most
// 2048 workgroups so the launch fills the 256 CUs without a long block queue.
//
// Reference kernels replaced: fp8_to_half_kernel (fp8_matmul.metal:215-223),
// float_to_fp8_kernel (:228-236); reference math restated: decode :19-40,
// encode :44-92.

#include "fp8mi_common.h"

namespace {

constexpr int kBlock = 128;  // (128 / 256 / 512 / 1024 threads: 2^30 fp32 encode 922 / 967 / 932 / 931 us, dequant 595 / 605 / 609 / 610 us)
// One pass per workgroup up to 2^30 threads (a grid-stride loop only beyond that): with the grid capped at 2048
// workgroups the 2^30-element casts ran at 4.6-4.7 TB/s, with one-shot workgroups (262,144 of them for the fp32
// encode) at 5.8 (encode) / 5.3 (dequant) TB/s - the dispatcher keeps the memory pipes fuller than a long loop does.
constexpr int kMaxGrid = 1 << 22;
constexpr int kDequantUnroll = 1;
#ifndef FP8MI_CAST_UNROLL
#define FP8MI_CAST_UNROLL 4  // 16-byte loads in flight per lane in the vector cast kernels
#endif

// ---------------------------------------------------------------------------
// decode: four packed fp8 bytes -> four halves (two packed dwords), exact.
// Bit trick: fp8 [s eeee mmm] placed at half bits [15 | 13:7] is the half
// 2^(e-15)(1+m/8) (or the half-subnormal m/8 * 2^-14 for e == 0); multiplying
// by 2^8 rebiases it to the e4m3 value 2^(e-7)(1+m/8) (m/8 * 2^-6).  NaN
// patterns become +0.0 first (fp8_matmul.metal:21).
// ---------------------------------------------------------------------------
FP8MI_DEVICE uint32_t fp8x2_to_half2_bits(uint32_t x /* bytes at [15:8] and [31:24] */)
{
    uint32_t t = x & 0x7F007F00u;
    uint32_t h = (x & 0x80008000u) | (t >> 1);
    uint32_t m = (t + 0x01000100u) & 0x80008000u;  // bit15 of a half set iff its byte is NaN
    m = m | (m - (m >> 15));                       // 0xFFFF per NaN half
    return h & ~m;
}

FP8MI_DEVICE void decode4_half(uint32_t w, f16x2 &lo, f16x2 &hi)
{
    uint32_t a = __builtin_amdgcn_perm(0u, w, 0x010c000cu);  // (b0 << 8) | (b1 << 24)
    uint32_t b = __builtin_amdgcn_perm(0u, w, 0x030c020cu);  // (b2 << 8) | (b3 << 24)
    const f16x2 k256 = {(_Float16)256.0f, (_Float16)256.0f};
    lo = __builtin_bit_cast(f16x2, fp8x2_to_half2_bits(a)) * k256;
    hi = __builtin_bit_cast(f16x2, fp8x2_to_half2_bits(b)) * k256;
}

// one byte -> half, same bit trick (a true f16 multiply keeps the sign of -0.0)
FP8MI_DEVICE _Float16 decode1_half(uint32_t b)
{
    const uint16_t hb = (uint16_t)fp8x2_to_half2_bits((b & 0xFFu) << 8);
    return __builtin_bit_cast(_Float16, hb) * (_Float16)256.0f;
}

template <int OUT>
struct OutVec;  // 16 output elements

template <>
struct OutVec<FP8MI_F16> {
    static FP8MI_DEVICE void store(void *out, int64_t i16, const f16x2 (&h)[8])
    {
        u32x4 *o = (u32x4 *)out + i16 * 2;
        u32x4 v0, v1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v0[j] = __builtin_bit_cast(uint32_t, h[j]);
            v1[j] = __builtin_bit_cast(uint32_t, h[4 + j]);
        }
        o[0] = v0;
        o[1] = v1;
    }
    static FP8MI_DEVICE void store1(void *out, int64_t i, _Float16 v) { ((_Float16 *)out)[i] = v; }
};

template <>
struct OutVec<FP8MI_F32> {
    static FP8MI_DEVICE void store(void *out, int64_t i16, const f16x2 (&h)[8])
    {
        f32x4 *o = (f32x4 *)out + i16 * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v = {(float)h[2 * j][0], (float)h[2 * j][1], (float)h[2 * j + 1][0], (float)h[2 * j + 1][1]};
            o[j] = v;
        }
    }
    static FP8MI_DEVICE void store1(void *out, int64_t i, _Float16 v) { ((float *)out)[i] = (float)v; }
};

template <>
struct OutVec<FP8MI_BF16> {
    static FP8MI_DEVICE uint32_t pack(f16x2 h)
    {
        __bf16 a = (__bf16)(float)h[0], b = (__bf16)(float)h[1];
        return (uint32_t)__builtin_bit_cast(uint16_t, a) | ((uint32_t)__builtin_bit_cast(uint16_t, b) << 16);
    }
    static FP8MI_DEVICE void store(void *out, int64_t i16, const f16x2 (&h)[8])
    {
        u32x4 *o = (u32x4 *)out + i16 * 2;
        u32x4 v0, v1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v0[j] = pack(h[j]);
            v1[j] = pack(h[4 + j]);
        }
        o[0] = v0;
        o[1] = v1;
    }
    static FP8MI_DEVICE void store1(void *out, int64_t i, _Float16 v) { ((__bf16 *)out)[i] = (__bf16)(float)v; }
};

// in/out 16-byte aligned: 16 fp8 bytes per lane per load (1 KiB per load
// instruction), the lane's 16 outputs stored as 16-byte pieces; 4 loads in
// flight per lane; n16 = count / 16 full vectors, then a scalar tail.
template <int OUT>
__global__ __launch_bounds__(kBlock) void dequant_kernel(const uint8_t *__restrict__ in, void *__restrict__ out,
                                                            const float *__restrict__ scale, int64_t count)
{
    const bool has_scale = scale != nullptr;
    _Float16 s = has_scale ? (_Float16)scale[0] : (_Float16)1.0f;
    const f16x2 s2 = {s, s};
    const int64_t n16 = count >> 4;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const u32x4 *in4 = (const u32x4 *)in;
    // ONE 16-byte piece per lane and pass (round 3, tools/probes/copy_sweep.hip: the memory instructions of this kernel alone - 16 B in, 2 x 16 B out
    // per lane - move 6.07 TB/s as one pass per workgroup and 5.28 TB/s in round 2's shape of 4 pieces per lane on a quarter of the workgroups)
    constexpr int kUn = kDequantUnroll;
    for (int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i0 < n16; i0 += stride * kUn) {
        u32x4 w[kUn];
#pragma unroll
        for (int u = 0; u < kUn; ++u) {
            const int64_t i = i0 + u * stride;
            w[u] = i < n16 ? __builtin_nontemporal_load(in4 + i) : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < kUn; ++u) {
            const int64_t i = i0 + u * stride;
            if (i >= n16) break;
            f16x2 h[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) decode4_half(w[u][j], h[2 * j], h[2 * j + 1]);
            if (has_scale) {
#pragma unroll
                for (int j = 0; j < 8; ++j) h[j] = h[j] * s2;
            }
            OutVec<OUT>::store(out, i, h);
        }
    }
    if (blockIdx.x == 0) {
        int64_t i = (n16 << 4) + threadIdx.x;
        if (i < count) {
            _Float16 v = decode1_half(in[i]);
            if (has_scale) v = v * s;
            OutVec<OUT>::store1(out, i, v);
        }
    }
}

// unaligned fallback: one element per lane per step
template <int OUT>
__global__ __launch_bounds__(kBlock) void dequant_scalar_kernel(const uint8_t *__restrict__ in, void *__restrict__ out,
                                                                 const float *__restrict__ scale, int64_t count)
{
    const bool has_scale = scale != nullptr;
    _Float16 s = has_scale ? (_Float16)scale[0] : (_Float16)1.0f;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += stride) {
        _Float16 v = decode1_half(in[i]);
        if (has_scale) v = v * s;
        OutVec<OUT>::store1(out, i, v);
    }
}

// ---------------------------------------------------------------------------
// encode: float32 bits -> fp8 byte, integer-only.
// ---------------------------------------------------------------------------

// reference semantics (fp8_matmul.metal:44-92, exact-arithmetic behaviour of
// its Python twin test_fp8_correctness.py:53-106)
FP8MI_DEVICE uint32_t encode_ref_bits(uint32_t bits)
{
    uint32_t a = bits & 0x7FFFFFFFu;
    uint32_t sign = (a != 0u) ? ((bits >> 24) & 0x80u) : 0u;  // `val < 0`: -0.0 has no sign (:46)
    uint32_t e = a >> 23, man = a & 0x7FFFFFu;
    // normal range: top three mantissa bits, RNE on the low 20, clamp (no carry, :79-81)
    uint32_t qn = (man + 0x7FFFFu + ((man >> 20) & 1u)) >> 20;
    qn = min(qn, 7u);
    uint32_t eb = e - 120u;
    qn = (eb == 15u && qn == 7u) ? 6u : qn;  // never the NaN pattern (:87-89)
    uint32_t normal = (eb << 3) | qn;
    // subnormal range [2^-9, 2^-6): mant = RNE(val * 512), clamp to 7 (:64-70)
    uint32_t s = 141u - e;  // 21..23 in this range
    s = min(max(s, 1u), 31u);
    uint32_t full = man | 0x800000u;
    uint32_t qs = (full + ((1u << (s - 1u)) - 1u) + ((full >> s) & 1u)) >> s;
    qs = min(qs, 7u);
    uint32_t r = (a < 0x3C800000u) ? qs : normal;  // < 2^-6
    r = (a < 0x3B000000u) ? 0u : r;                 // < 2^-9 flushes, sign kept (:58-60)
    r = (a >= 0x43E00000u) ? 0x7Eu : r;             // >= 448 saturates, also inf (:53-55)
    r |= sign;
    return (a > 0x7F800000u) ? 0x7Fu : r;           // NaN in: outside the reference's domain
}

// OCP e4m3fn round-to-nearest-even with overflow to NaN: what torch-CPU
// `.to(torch.float8_e4m3fn)` produces (non-default mode).
FP8MI_DEVICE uint32_t encode_rne_bits(uint32_t bits)
{
    uint32_t a = bits & 0x7FFFFFFFu;
    uint32_t sign = (bits >> 24) & 0x80u;
    uint32_t e = a >> 23, man = a & 0x7FFFFFu;
    uint32_t v = ((e - 120u) << 23) | man;
    uint32_t n = (v + 0x7FFFFu + ((v >> 20) & 1u)) >> 20;  // carry may bump the exponent
    uint32_t s = min(max(141u - e, 1u), 31u);
    uint32_t full = man | 0x800000u;
    uint32_t qs = (full + ((1u << (s - 1u)) - 1u) + ((full >> s) & 1u)) >> s;
    qs = (e < 110u) ? 0u : qs;  // below 2^-17: far under half the smallest subnormal
    uint32_t r = (e < 121u) ? qs : n;
    r = (r > 0x7Eu) ? 0x7Fu : r;
    r = (a > 0x7F800000u) ? 0x7Fu : r;
    return r | sign;
}

template <int MODE>
FP8MI_DEVICE uint32_t encode_bits(float v)
{
    uint32_t b = __float_as_uint(v);
    return MODE == FP8MI_ENC_REFERENCE ? encode_ref_bits(b) : encode_rne_bits(b);
}

// Reference-semantics encode of TWO floats around the hardware convert.
// v_cvt_pk_fp8_f32 does the in-range work (OCP round-to-nearest-even onto the
// e4m3 grid, subnormals included); what the reference does differently
// (fp8_matmul.metal:44-92) is patched with a few integer ops per element:
//   * no carry: if rounding bumped the exponent field (1.9375 -> 2.0, or the
//     top subnormal -> 2^-6) the reference keeps mantissa 7 of the ORIGINAL
//     binade, which is exactly "one code below" the carried result;
//   * |x| < 2^-9 flushes to zero (sign kept), |x| >= 448 and inf saturate to
//     0x7E - both selected explicitly, so the hardware's own underflow /
//     overflow behaviour is never relied upon;
//   * -0.0 -> 0x00; NaN (outside the reference's domain) -> 0x7F.
// 19 VALU ops per element instead of 37 for the all-integer form; verified on
// the same 147k golden vectors.
FP8MI_DEVICE uint32_t encode_ref_pair(float v0, float v1)
{
    const uint32_t b0 = __float_as_uint(v0), b1 = __float_as_uint(v1);
    const uint32_t a0 = b0 & 0x7FFFFFFFu, a1 = b1 & 0x7FFFFFFFu;
    const uint32_t pk = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(a0), __uint_as_float(a1), 0, false);
    uint32_t h0 = pk & 0xFFu, h1 = (pk >> 8) & 0xFFu;
    // exponent field the reference keeps: that of the input's own binade (0 below 2^-6)
    const uint32_t e0 = max(a0 >> 23, 120u) - 120u, e1 = max(a1 >> 23, 120u) - 120u;
    h0 -= ((h0 >> 3) != e0) ? 1u : 0u;
    h1 -= ((h1 >> 3) != e1) ? 1u : 0u;
    h0 = (a0 < 0x3B000000u) ? 0u : h0;
    h1 = (a1 < 0x3B000000u) ? 0u : h1;
    h0 = (a0 >= 0x43E00000u) ? 0x7Eu : h0;
    h1 = (a1 >= 0x43E00000u) ? 0x7Eu : h1;
    h0 |= (a0 != 0u) ? ((b0 >> 24) & 0x80u) : 0u;
    h1 |= (a1 != 0u) ? ((b1 >> 24) & 0x80u) : 0u;
    h0 = (a0 > 0x7F800000u) ? 0x7Fu : h0;
    h1 = (a1 > 0x7F800000u) ? 0x7Fu : h1;
    return h0 | (h1 << 8);
}

// torch / OCP semantics (FP8MI_ENC_RNE) of TWO floats around the same hardware convert: gfx950's v_cvt_pk_fp8_f32 IS
// OCP round-to-nearest-even onto the e4m3fn grid, so inside [2^-9, 448] its byte is torch-CPU's byte; the edges are
// selected explicitly so that the instruction's own underflow / overflow behaviour is never relied upon:
//   |x| <= 2^-10 -> 0 (the tie with the smallest subnormal goes to even), 2^-10 < |x| < 2^-9 -> 0x01,
//   448 < |x| <= 464 -> 0x7E (464 ties to even), |x| > 464, inf and NaN -> 0x7F; the sign bit is the input's (-0.0 -> 0x80).
// ~12 VALU ops per element instead of ~37 for encode_rne_bits (kept for the scalar tails); byte-exact against
// torch-CPU `.to(float8_e4m3fn)` on the 147k golden vectors (tests/test_gpu_parity.py, the check of test_mps_vs_cpu.py:283-357).
FP8MI_DEVICE uint32_t encode_rne_pair(float v0, float v1)
{
    const uint32_t b0 = __float_as_uint(v0), b1 = __float_as_uint(v1);
    const uint32_t a0 = b0 & 0x7FFFFFFFu, a1 = b1 & 0x7FFFFFFFu;
    const uint32_t pk = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(__uint_as_float(a0), __uint_as_float(a1), 0, false);
    uint32_t h0 = pk & 0xFFu, h1 = (pk >> 8) & 0xFFu;
    h0 = (a0 < 0x3B000000u) ? 1u : h0;
    h1 = (a1 < 0x3B000000u) ? 1u : h1;
    h0 = (a0 <= 0x3A800000u) ? 0u : h0;
    h1 = (a1 <= 0x3A800000u) ? 0u : h1;
    h0 = (a0 > 0x43E00000u) ? 0x7Eu : h0;
    h1 = (a1 > 0x43E00000u) ? 0x7Eu : h1;
    h0 = (a0 > 0x43E80000u) ? 0x7Fu : h0;   // also inf and NaN
    h1 = (a1 > 0x43E80000u) ? 0x7Fu : h1;
    h0 |= (b0 >> 24) & 0x80u;
    h1 |= (b1 >> 24) & 0x80u;
    return h0 | (h1 << 8);
}

// four floats -> four packed bytes
template <int MODE>
FP8MI_DEVICE uint32_t encode4(float v0, float v1, float v2, float v3)
{
    if (MODE == FP8MI_ENC_REFERENCE) return encode_ref_pair(v0, v1) | (encode_ref_pair(v2, v3) << 16);
    return encode_rne_pair(v0, v1) | (encode_rne_pair(v2, v3) << 16);
}

template <int IN>
struct InVec;  // loads 16 elements as float

template <>
struct InVec<FP8MI_F32> {
    static FP8MI_DEVICE void load(const void *in, int64_t i16, float (&f)[16])
    {
        const f32x4 *p = (const f32x4 *)in + i16 * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 v = __builtin_nontemporal_load(p + j);
            f[4 * j] = v[0]; f[4 * j + 1] = v[1]; f[4 * j + 2] = v[2]; f[4 * j + 3] = v[3];
        }
    }
    static FP8MI_DEVICE float load1(const void *in, int64_t i) { return ((const float *)in)[i]; }
    static constexpr int kPer = 4;  // elements per 16-byte load
    static FP8MI_DEVICE void loadv(const void *in, int64_t iv, float (&f)[8])
    {
        f32x4 v = __builtin_nontemporal_load((const f32x4 *)in + iv);
        f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
    }
};

template <>
struct InVec<FP8MI_F16> {
    static FP8MI_DEVICE void load(const void *in, int64_t i16, float (&f)[16])
    {
        const u32x4 *p = (const u32x4 *)in + i16 * 2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            u32x4 v = __builtin_nontemporal_load(p + j);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // NB: bit_cast straight from the vector element v[q] miscompiles
                // (every q reads element 0, hipcc 7.2); go through a scalar.
                const uint32_t w = v[q];
                f[8 * j + 2 * q] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xFFFFu));
                f[8 * j + 2 * q + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
            }
        }
    }
    static FP8MI_DEVICE float load1(const void *in, int64_t i) { return (float)((const _Float16 *)in)[i]; }
    static constexpr int kPer = 8;
    static FP8MI_DEVICE void loadv(const void *in, int64_t iv, float (&f)[8])
    {
        u32x4 v = __builtin_nontemporal_load((const u32x4 *)in + iv);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t w = v[q];  // (bit_cast straight from v[q] miscompiles, see load())
            f[2 * q] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xFFFFu));
            f[2 * q + 1] = (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16));
        }
    }
};

template <>
struct InVec<FP8MI_BF16> {
    static FP8MI_DEVICE void load(const void *in, int64_t i16, float (&f)[16])
    {
        const u32x4 *p = (const u32x4 *)in + i16 * 2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            u32x4 v = __builtin_nontemporal_load(p + j);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f[8 * j + 2 * q] = __uint_as_float(v[q] << 16);
                f[8 * j + 2 * q + 1] = __uint_as_float(v[q] & 0xFFFF0000u);
            }
        }
    }
    static FP8MI_DEVICE float load1(const void *in, int64_t i)
    {
        return __uint_as_float((uint32_t)((const uint16_t *)in)[i] << 16);
    }
    static constexpr int kPer = 8;
    static FP8MI_DEVICE void loadv(const void *in, int64_t iv, float (&f)[8])
    {
        u32x4 v = __builtin_nontemporal_load((const u32x4 *)in + iv);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f[2 * q] = __uint_as_float(v[q] << 16);
            f[2 * q + 1] = __uint_as_float(v[q] & 0xFFFF0000u);
        }
    }
};

// in/out 16-byte aligned.  Lane l loads the 16 bytes at 16 l of each KiB piece of
// the input (one contiguous KiB per load instruction) = kPer elements, and
// stores their kPer bytes at kPer * l (256 / 512 contiguous bytes per store
// instruction); 4 pieces per lane are in flight.
// scale = 448 / amax evaluated in double, as the reference's Python float arithmetic
// (fp8_mps_native.py:174-176); amax == 0 -> 1
FP8MI_DEVICE float scale_from_amax(float amax) { return amax > 0.0f ? (float)(448.0 / (double)amax) : 1.0f; }

// FROM_AMAX: `prescale` points at {amax, inv_scale}: every thread derives the scale from
// amax (read-only here), thread 0 of workgroup 0 publishes float(1 / scale) in slot 1
// (fp8_mps_native.py:189) - the amax-scaled quantiser without a separate scale kernel.
template <int IN>
constexpr int kEncodeUnroll = IN == FP8MI_F32 ? 1 : 2;

template <int IN, int MODE, bool FROM_AMAX = false>
__global__ __launch_bounds__(kBlock) void encode_kernel(const void *__restrict__ in, uint8_t *__restrict__ out,
                                                         const float *__restrict__ prescale, int64_t count)
{
    // vectors in flight per lane (the grid covers 16 elements per lane): measured at 2^30 fp32 elements
    // 1 / 2 / 4 / 8 = 6.1 / 5.9 / 5.7 / 5.5 TB/s, at 50 M bf16 elements 3.8 / 4.3 / 4.2 / 3.6
    constexpr int kPer = InVec<IN>::kPer, kUn = kEncodeUnroll<IN>;
    const bool has_ps = prescale != nullptr;
    float ps = has_ps ? prescale[0] : 1.0f;
    if (FROM_AMAX) {
        const float amax = ps;
        ps = scale_from_amax(amax);
        if (blockIdx.x == 0 && threadIdx.x == 0)
            ((float *)prescale)[1] = amax > 0.0f ? (float)(1.0 / (448.0 / (double)amax)) : 1.0f;
    }
    const int64_t nv = count / kPer;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i0 < nv; i0 += stride * kUn) {
        float f[kUn][8];
#pragma unroll
        for (int u = 0; u < kUn; ++u) {
            const int64_t i = i0 + u * stride;
            if (i < nv) InVec<IN>::loadv(in, i, f[u]);
        }
#pragma unroll
        for (int u = 0; u < kUn; ++u) {
            const int64_t i = i0 + u * stride;
            if (i >= nv) break;
            if (has_ps) {
#pragma unroll
                for (int j = 0; j < kPer; ++j) f[u][j] = f[u][j] * ps;  // float32 multiply, as `inp * scale` (fp8_mps_native.py:179)
            }
            const uint32_t w0 = encode4<MODE>(f[u][0], f[u][1], f[u][2], f[u][3]);
            // streaming stores: the bytes are written once and not read back here (copy_sweep.hip, this kernel's memory shape: 6.63 TB/s
            // with nt stores against 6.33 with the default policy)
            if (kPer == 4) {
                __builtin_nontemporal_store(w0, (uint32_t *)out + i);
            } else {
                const uint32_t w1 = encode4<MODE>(f[u][4], f[u][5], f[u][6], f[u][7]);
                __builtin_nontemporal_store(u32x2{w0, w1}, (u32x2 *)out + i);
            }
        }
    }
    if (blockIdx.x == 0) {
        int64_t i = nv * kPer + threadIdx.x;
        if (i < count) {
            float v = InVec<IN>::load1(in, i);
            if (has_ps) v = v * ps;
            out[i] = (uint8_t)encode_bits<MODE>(v);
        }
    }
}

template <int IN, int MODE, bool FROM_AMAX = false>
__global__ __launch_bounds__(kBlock) void encode_scalar_kernel(const void *__restrict__ in, uint8_t *__restrict__ out,
                                                                const float *__restrict__ prescale, int64_t count)
{
    const bool has_ps = prescale != nullptr;
    float ps = has_ps ? prescale[0] : 1.0f;
    if (FROM_AMAX) {
        const float amax = ps;
        ps = scale_from_amax(amax);
        if (blockIdx.x == 0 && threadIdx.x == 0)
            ((float *)prescale)[1] = amax > 0.0f ? (float)(1.0 / (448.0 / (double)amax)) : 1.0f;
    }
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += stride) {
        float v = InVec<IN>::load1(in, i);
        if (has_ps) v = v * ps;
        out[i] = (uint8_t)encode_bits<MODE>(v);
    }
}

// ---------------------------------------------------------------------------
// amax: |x| max as float bits through atomicMax on the (non-negative) pattern
// ---------------------------------------------------------------------------
// amax runs few, fat workgroups (1024 threads, <= 512 of them): every workgroup ends in one same-address atomic,
// and those serialise at ~10 ns each - 1536 workgroups finishing together cost ~15 us on a 25 MB tensor
constexpr int kAmaxBlock = 1024;
constexpr int kAmaxMaxGrid = 512;

template <int IN>
__global__ __launch_bounds__(kAmaxBlock) void amax_kernel(const void *__restrict__ in, uint32_t *__restrict__ out_bits,
                                                       int64_t count, int vec_ok)
{
    // like the encode kernel: every wave-instruction reads one contiguous KiB (lane l -> 16 B at 16 l), four of them
    // in flight per lane (the first version read 16 consecutive elements per lane: 25-50 % of each line per
    // instruction and one batch in flight - 0.9 TB/s on a 25 MB activation tensor)
    constexpr int P = InVec<IN>::kPer;   // elements per 16-byte vector
    constexpr int U = FP8MI_CAST_UNROLL;
    float m = 0.0f;
    const int64_t stride = (int64_t)gridDim.x * kAmaxBlock;
    const int64_t nv = vec_ok ? count / P : 0;
    int64_t i = (int64_t)blockIdx.x * kAmaxBlock + threadIdx.x;
    for (; i + (U - 1) * stride < nv; i += U * stride) {
        float f[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) InVec<IN>::loadv(in, i + u * stride, f[u]);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < P; ++j) m = fmaxf(m, fabsf(f[u][j]));  // fmaxf drops NaN operands
    }
    for (; i < nv; i += stride) {
        float f[8];
        InVec<IN>::loadv(in, i, f);
#pragma unroll
        for (int j = 0; j < P; ++j) m = fmaxf(m, fabsf(f[j]));
    }
    for (int64_t e = nv * P + (int64_t)blockIdx.x * kAmaxBlock + threadIdx.x; e < count; e += stride)
        m = fmaxf(m, fabsf(InVec<IN>::load1(in, e)));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    __shared__ float wmax[kAmaxBlock / 64];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float b = wmax[0];
#pragma unroll
        for (int w = 1; w < kAmaxBlock / 64; ++w) b = fmaxf(b, wmax[w]);
        // Same-address atomics serialise at the memory side (~10 ns each): 2048 workgroups finishing together cost
        // ~20 us on a 25 MB tensor.  The running maximum only grows, so a workgroup whose maximum does not exceed
        // the value it can already see has nothing to add (a stale, smaller value merely costs the atomic).
        const uint32_t mine = __float_as_uint(b);
        if (mine > __hip_atomic_load(out_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out_bits, mine);
    }
}

int grid_for(int64_t work_items)
{
    int64_t g = (work_items + kBlock - 1) / kBlock;
    if (g < 1) g = 1;
    if (g > kMaxGrid) g = kMaxGrid;
    return (int)g;
}

bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

}  // namespace

// ---------------------------------------------------------------------------
// host launchers (called from fp8mi_api.hip)
// ---------------------------------------------------------------------------
int fp8mi_launch_dequant(const uint8_t *in, void *out, const float *scale, int64_t count, int out_dtype,
                         hipStream_t s)
{
    if (count == 0) return 0;
    const bool vec = aligned16(in) && aligned16(out);
    const int grid = grid_for(vec ? (count >> 4) / kDequantUnroll : count);  // vector kernel: 16 bytes per lane per pass
#define FP8MI_DQ(OUT)                                                                                          \
    (vec ? fp8mi_launch(dequant_kernel<OUT>, dim3(grid), dim3(kBlock), s, in, out, scale, count)               \
         : fp8mi_launch(dequant_scalar_kernel<OUT>, dim3(grid), dim3(kBlock), s, in, out, scale, count))
    if (out_dtype == FP8MI_F16) return FP8MI_DQ(FP8MI_F16);
    if (out_dtype == FP8MI_F32) return FP8MI_DQ(FP8MI_F32);
    return FP8MI_DQ(FP8MI_BF16);
#undef FP8MI_DQ
}

template <int IN, bool FROM_AMAX = false>
static int launch_encode_in(const void *in, uint8_t *out, const float *prescale, int64_t count, int mode,
                            hipStream_t s)
{
    const bool vec = aligned16(in) && aligned16(out);
    // 16 elements per lane: 16-bit inputs in one pass of 2 vectors, fp32 in 4 passes of one (measured best: a million
    // one-vector workgroups instead reach only 5.1 TB/s)
    const int grid = grid_for(vec ? (count >> 4) + 1 : count);
    if (mode == FP8MI_ENC_REFERENCE) {
        if (vec) return fp8mi_launch(encode_kernel<IN, FP8MI_ENC_REFERENCE, FROM_AMAX>, dim3(grid), dim3(kBlock), s, in, out, prescale, count);
        return fp8mi_launch(encode_scalar_kernel<IN, FP8MI_ENC_REFERENCE, FROM_AMAX>, dim3(grid), dim3(kBlock), s, in, out, prescale, count);
    }
    if (vec) return fp8mi_launch(encode_kernel<IN, FP8MI_ENC_RNE, FROM_AMAX>, dim3(grid), dim3(kBlock), s, in, out, prescale, count);
    return fp8mi_launch(encode_scalar_kernel<IN, FP8MI_ENC_RNE, FROM_AMAX>, dim3(grid), dim3(kBlock), s, in, out, prescale, count);
}

int fp8mi_launch_encode(const void *in, int in_dtype, uint8_t *out, const float *prescale, int64_t count, int mode,
                        hipStream_t s)
{
    if (count == 0) return 0;
    if (in_dtype == FP8MI_F32) return launch_encode_in<FP8MI_F32>(in, out, prescale, count, mode, s);
    if (in_dtype == FP8MI_F16) return launch_encode_in<FP8MI_F16>(in, out, prescale, count, mode, s);
    return launch_encode_in<FP8MI_BF16>(in, out, prescale, count, mode, s);
}

int fp8mi_launch_amax(const void *in, int in_dtype, float *out, int64_t count, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    if (count == 0) return 0;
    const int vec = aligned16(in) ? 1 : 0;
    // one 16-byte vector per lane and iteration, FP8MI_CAST_UNROLL of them in flight
    const int per = in_dtype == FP8MI_F32 ? 4 : 8;
    const int64_t items = vec ? (count / per + FP8MI_CAST_UNROLL - 1) / FP8MI_CAST_UNROLL + 1 : count;
    int64_t grid = (items + kAmaxBlock - 1) / kAmaxBlock;
    grid = grid < 1 ? 1 : (grid > kAmaxMaxGrid ? kAmaxMaxGrid : grid);
    uint32_t *ob = (uint32_t *)out;
    if (in_dtype == FP8MI_F32) return fp8mi_launch(amax_kernel<FP8MI_F32>, dim3((unsigned)grid), dim3(kAmaxBlock), s, in, ob, count, vec);
    if (in_dtype == FP8MI_F16) return fp8mi_launch(amax_kernel<FP8MI_F16>, dim3((unsigned)grid), dim3(kAmaxBlock), s, in, ob, count, vec);
    return fp8mi_launch(amax_kernel<FP8MI_BF16>, dim3((unsigned)grid), dim3(kAmaxBlock), s, in, ob, count, vec);
}

int fp8mi_launch_quantize(const void *in, int in_dtype, uint8_t *out, float *scales, int64_t count, int mode,
                          hipStream_t s)
{
    // scales[0] <- amax (atomicMax over the input), then one encode launch that derives
    // scale = 448 / amax per thread and publishes scales[1] = 1 / scale
    int rc = fp8mi_launch_amax(in, in_dtype, scales, count, s);
    if (rc) return rc;
    // (count == 0 still launches one workgroup: it publishes inv_scale = 1 and touches no data)
    if (in_dtype == FP8MI_F32) return launch_encode_in<FP8MI_F32, true>(in, out, scales, count, mode, s);
    if (in_dtype == FP8MI_F16) return launch_encode_in<FP8MI_F16, true>(in, out, scales, count, mode, s);
    return launch_encode_in<FP8MI_BF16, true>(in, out, scales, count, mode, s);
}
