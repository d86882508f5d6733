// Shared device helpers for the fp8mi kernels (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/fp8mi.h"

typedef float    f32x2  __attribute__((ext_vector_type(2)));
typedef float    f32x4  __attribute__((ext_vector_type(4)));
typedef float    f32x16 __attribute__((ext_vector_type(16)));
typedef int      i32x4  __attribute__((ext_vector_type(4)));
typedef int      i32x8  __attribute__((ext_vector_type(8)));
typedef uint32_t u32x2  __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4  __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2  __attribute__((ext_vector_type(2)));

#define FP8MI_DEVICE __device__ __forceinline__

// ---------------------------------------------------------------------------
// epilogue parameters shared by the GEMV / GEMM / generic kernels
// ---------------------------------------------------------------------------
struct MMParams {
    const uint8_t *A;      // (M,K) row-major bytes
    const uint8_t *B;      // (N,K) row-major bytes
    void *C;               // (M,N) out_dtype
    const float *scale_a;  // [1] or [M]
    const float *scale_b;  // [1] or [N]
    const void *bias;      // nullptr or [N] bias_dtype
    const float *scale_result;  // nullptr or [1]
    int64_t M, N, K;
    int64_t lda, ldb, ldc;
    int sa_row, sb_row;    // 0 per-tensor, 1 per-row
    int out_dtype, bias_dtype;
    int transposed;        // 1: the caller computes C^T = B . A^T (sharded linear): bias runs along M (bias[m]) and the epilogue
                           //    multiplies by scale_b first, then scale_a - the bits of the untransposed fused epilogue
    int nan_zero;          // 1: NaN bytes decode to 0 (reference), 0: propagate
    int debug;             // diagnostic builds only (FP8MI_STAMP): ablation bits, else 0
    int split;             // GEMM tile kernels: number of K ranges per tile (1 = no split-K)
    uint8_t *ws;           // split-K workspace (tile counters, then fp32 partial tiles) or nullptr
    int64_t ws_bytes;
};

constexpr int kWsCounterBytes = 4096;  // 1024 tile counters, zero between launches

// Kernel arguments arrive through the kernarg segment, which the runtime may keep in HOST memory (a scalar load from it
// is a PCIe round trip, ~1.5 us; HIP_FORCE_DEV_KERNARG=1 moves it to HBM, ~0.5 us).  Left alone, hipcc loads the fields
// lazily - the GEMM had four dependent load-and-wait groups ahead of its first DMA and another in the epilogue, 4.6 us
// from kernel entry to the K loop (in-kernel stamps, profiles/r02_stamps.txt).  pin_params() makes every field opaque at
// kernel entry: all scalar loads issue as ONE clause with one wait, and the values then live in SGPRs (the compiler
// cannot re-derive them from the kernarg pointer later).
#define FP8MI_PIN_S(x) asm volatile("" : "+s"(x))
FP8MI_DEVICE MMParams pin_params(const MMParams &p)
{
    MMParams q = p;
    FP8MI_PIN_S(q.A); FP8MI_PIN_S(q.B); FP8MI_PIN_S(q.C); FP8MI_PIN_S(q.scale_a); FP8MI_PIN_S(q.scale_b);
    FP8MI_PIN_S(q.bias); FP8MI_PIN_S(q.scale_result);
    FP8MI_PIN_S(q.M); FP8MI_PIN_S(q.N); FP8MI_PIN_S(q.K); FP8MI_PIN_S(q.lda); FP8MI_PIN_S(q.ldb); FP8MI_PIN_S(q.ldc);
    FP8MI_PIN_S(q.sa_row); FP8MI_PIN_S(q.sb_row); FP8MI_PIN_S(q.out_dtype); FP8MI_PIN_S(q.bias_dtype);
    FP8MI_PIN_S(q.transposed); FP8MI_PIN_S(q.nan_zero); FP8MI_PIN_S(q.split); FP8MI_PIN_S(q.ws); FP8MI_PIN_S(q.ws_bytes);
    return q;
}

// Cache policy of the GEMM kernels' C stores (compile-time knob for A/B builds; the product value is the default below):
//   0 default (write-back)   1 nt (streaming)   2 sc0 sc1 (system scope: written THROUGH the XCD's L2)   3 sc0 sc1 nt   4 sc1   5 sc1 nt
// The end of a kernel is an agent-scope release, which on this 8-L2 part means a write-back of every dirty line of
// every L2: stores that are written through leave nothing for it.
#ifndef FP8MI_CSTORE
#define FP8MI_CSTORE 1
#endif
constexpr int kCStoreAux = FP8MI_CSTORE == 0 ? 0 : FP8MI_CSTORE == 1 ? 2 : FP8MI_CSTORE == 2 ? 17 : FP8MI_CSTORE == 3 ? 19 : FP8MI_CSTORE == 4 ? 16 : 18;
FP8MI_DEVICE void store_c16(u32x4 q, void *ptr)   // one 16-byte piece of C
{
#if FP8MI_CSTORE == 0
    *(u32x4 *)ptr = q;
#elif FP8MI_CSTORE == 1
    __builtin_nontemporal_store(q, (u32x4 *)ptr);
#elif FP8MI_CSTORE == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(ptr), "v"(q) : "memory");
#elif FP8MI_CSTORE == 3
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(ptr), "v"(q) : "memory");
#elif FP8MI_CSTORE == 4
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(q) : "memory");
#else
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(ptr), "v"(q) : "memory");
#endif
}

// SWAR scrub: zero every byte of w whose low 7 bits are all ones (the two
// e4m3fn NaN patterns 0x7F / 0xFF), i.e. the reference's decode rule
// "NaN -> 0.0" (fp8_matmul.metal:21) applied to four packed bytes.
FP8MI_DEVICE uint32_t scrub_nan4(uint32_t w)
{
    uint32_t n = ~(w | 0x80808080u);            // NaN byte -> 0x00, else 0x01..0x7F
    uint32_t m = ~((n + 0x7F7F7F7Fu) | 0x7F7F7F7Fu);  // 0x80 where the byte of n is zero
    m = m | (m - (m >> 7));                     // 0xFF where NaN
    return w & ~m;
}

// true if any of the four bytes is a NaN pattern
FP8MI_DEVICE bool has_nan4(uint32_t w)
{
    uint32_t n = ~(w | 0x80808080u);
    return (~((n + 0x7F7F7F7Fu) | 0x7F7F7F7Fu)) != 0u;
}

// one byte -> float, reference semantics (fp8_matmul.metal:19-40): used by the
// slow paths only; the hot kernels decode with v_cvt_pk_f32_fp8 / MFMA.
// Built from bits, with no float multiply on the signed value: hipcc fuses
// `x * 2^k` followed by a conversion into v_fma_mix*(x, 2^k, +0), and
// (-0 * c) + (+0) is +0 - the sign of 0x80 (-0.0, metal:39) would be lost.
FP8MI_DEVICE float decode_ref(uint32_t b)
{
    const uint32_t mag = b & 0x7Fu;
    if (mag == 0x7Fu) return 0.0f;
    const uint32_t sign = (b & 0x80u) << 24;
    uint32_t bits;
    if (mag >= 8u) bits = (mag << 20) + (120u << 23);           // normal: rebias 7 -> 127
    else bits = __float_as_uint((float)mag * 0x1p-9f);           // subnormal: m * 2^-9 (>= 0, exact)
    return __uint_as_float(bits | sign);
}

FP8MI_DEVICE float load_as_float(const void *p, int64_t i, int dtype)
{
    if (dtype == FP8MI_F32) return ((const float *)p)[i];
    if (dtype == FP8MI_F16) return (float)((const _Float16 *)p)[i];
    return (float)((const __bf16 *)p)[i];
}

FP8MI_DEVICE void store_from_float(void *p, int64_t i, float v, int dtype)
{
    if (dtype == FP8MI_F32) ((float *)p)[i] = v;
    else if (dtype == FP8MI_F16) ((_Float16 *)p)[i] = (_Float16)v;
    else ((__bf16 *)p)[i] = (__bf16)v;
}

// The reference epilogue, in its order (fp8_matmul.metal:144-146 then
// fp8_mps_patch.py:94-104): (sum * sa) * sb, + bias, * scale_result, cast.
FP8MI_DEVICE float epilogue_value(float sum, float sa, float sb, bool has_bias, float bias,
                                  bool has_sr, float sr, bool transposed = false)
{
    float r = transposed ? (sum * sb) * sa : (sum * sa) * sb;
    if (has_bias) r = r + bias;
    if (has_sr) r = r * sr;
    return r;
}

// wave64 all-lanes sum.  Four DPP adds inside each row of 16 lanes (quad swaps, then the two mirrors: every lane of a row ends with
// the row's sum), then the four row sums are read out as scalars.  (The generic `__shfl_xor` butterfly compiles to six DEPENDENT
// ds_bpermute_b32 + s_waitcnt pairs - ~100 cycles each - per reduced value: 24 of them in series sat in the tail of every workgroup
// of the 4-row vec-mat, ~1 us of config C1's 5.2 us.)
template <int CTRL>
FP8MI_DEVICE float dpp_add(float x)   // x + (x of the lane the DPP control selects; 0 where that lane is inactive)
{
    const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false);
    return x + __builtin_bit_cast(float, y);
}

// sum over aligned groups of P = 2 / 4 / 8 / 16 adjacent lanes (every lane of a group ends with the group's sum)
template <int P>
FP8MI_DEVICE float group_sum(float v)
{
    static_assert(P == 1 || P == 2 || P == 4 || P == 8 || P == 16, "a power of two inside a row of 16 lanes");
    if (P >= 2) v = dpp_add<0xB1>(v);    // quad_perm [1, 0, 3, 2]
    if (P >= 4) v = dpp_add<0x4E>(v);    // quad_perm [2, 3, 0, 1]
    if (P >= 8) v = dpp_add<0x141>(v);   // row_half_mirror: lane i <- lane 7 - i of its half row
    if (P >= 16) v = dpp_add<0x140>(v);  // row_mirror: lane i <- lane 15 - i of its row
    return v;
}

FP8MI_DEVICE float wave_sum(float v)
{
    v = group_sum<16>(v);
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16)),
                r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}

// Per-dispatch timing hook (fp8mi_profile_begin / _end in the C ABI): while a profile is open on the calling thread
// every kernel launch carries its own start/stop event pair, filled from the dispatch packet's timestamps - the same
// clock rocprofv3 --kernel-trace reads.  Outside a profile this is a plain launch.
bool fp8mi_next_profile_events(hipEvent_t *start, hipEvent_t *stop);

// Launch with the status of THIS launch (hipGetLastError() would report - and clear - any earlier error on the thread).
template <typename... KArgs>
int fp8mi_launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, hipStream_t s, KArgs... args)
{
    void *argv[] = {(void *)&args...};
    hipEvent_t e0, e1;
    if (fp8mi_next_profile_events(&e0, &e1))
        return (int)hipExtLaunchKernel((const void *)kernel, grid, block, argv, 0, s, e0, e1, 0);
    return (int)hipLaunchKernel((const void *)kernel, grid, block, argv, 0, s);
}

// compute units of the current device (cached per device; the dispatch heuristics scale with it)
int fp8mi_cu_count();

// launchers implemented in the .hip files (host side, internal linkage by name)
int fp8mi_launch_gemv(const MMParams &p, bool fp32_only, hipStream_t s);
int fp8mi_launch_gemv_variant(const MMParams &p, int id, hipStream_t s);  // diagnostic library only
bool fp8mi_gemv_supported(const MMParams &p);
int fp8mi_launch_gemv_mx(const MMParams &p, hipStream_t s);   // 2 <= M <= 8, the vec-mat's structure
bool fp8mi_gemv_mx_supported(const MMParams &p);
int fp8mi_launch_gemv_mx_variant(const MMParams &p, int id, hipStream_t s);  // diagnostic library only
int fp8mi_launch_gemm(const MMParams &p, int variant, hipStream_t s);
int fp8mi_choose_gemm_variant(const MMParams &p);   // the tile kernel AUTO picks (host-only)
bool fp8mi_gemm_supported(const MMParams &p);
int fp8mi_launch_gemm256(const MMParams &p, int variant, hipStream_t s);   // fp8mi_gemm256.hip: whole 256x256 tiles, hand-scheduled loop
bool fp8mi_gemm256_supported(const MMParams &p);
int fp8mi_launch_generic(const MMParams &p, hipStream_t s);
int fp8mi_launch_skinny(const MMParams &p, hipStream_t s);
int fp8mi_launch_gemm_pc(const MMParams &p, int variant, hipStream_t s);  // diagnostic library only
bool fp8mi_skinny_supported(const MMParams &p);
