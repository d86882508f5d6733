// 2 <= M <= 64 ("skinny") FP8 scaled matmul for gfx950: weight-streaming, MFMA.
//
// The reference sends M <= 16 to its one-thread-per-output kernel
// (fp8_mps_native.py:208, fp8_matmul.metal:99-147); batch 4 is one of its three
// published shapes.  At small M the work is reading W (N x K bytes) once, so the
// kernel is built like the GEMV, not like the tiled GEMM:
//   * a workgroup of KW waves owns 16 consecutive rows of W (one MFMA n-fragment); the waves split K
//     round-robin in 128-byte steps and reduce through LDS.  KW = 8 for deep K; for shallow K fewer, so that
//     every wave still has kU steps to keep in flight (K = 4096 with 8 waves left each wave 4 of its 8 load
//     slots: 25 us on a 56 MiB weight matrix);
//   * W fragments go HBM -> VGPR directly (non-temporal, no LDS: nothing is
//     shared between waves), U steps (2U loads of 1 KiB) in flight per wave;
//     like the GEMV the kernel is latency-, not issue-bound, so depth matters;
//   * x (M x K, <= 64 rows) is L2-resident; its fragments are loaded per step;
//   * one v_mfma_scale_f32_16x16x128_f8f6f4 per 16 rows of x per step - the
//     matrix core replaces 2 x 16 x 128 decode + FMA operations;
//   * NaN bytes: memory-bound, so every fragment is simply scrubbed (SWAR) when
//     the reference semantics are requested; fused epilogue as everywhere.

#include "fp8mi_common.h"

namespace {

constexpr int kScaleOne = 0x7F7F7F7F;

template <int TM, int kU /* K-steps per wave in flight */, int KW /* waves per workgroup = K shares of its fragment */>
__global__ __launch_bounds__(KW * 64) void skinny_kernel(MMParams p_in)
{
    const MMParams p = pin_params(p_in);  // every kernel argument in one scalar-load clause (fp8mi_common.h)
    __shared__ f32x4 part[KW][TM][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int kw = wave;  // K share
    const int64_t n0 = (int64_t)blockIdx.x * 16;
    const int64_t K = p.K;
    const int nk = (int)((K + 127) / 128);
    const bool wrow_ok = n0 + r < p.N;
    const uint8_t *wrow = p.B + (n0 + r) * p.ldb;
    const int c1 = g * 16, c2 = 64 + g * 16;  // this lane's two 16-byte chunks of a 128-byte K-step

    f32x4 acc[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    for (int s0 = kw; s0 < nk; s0 += KW * kU) {
        i32x8 wf[kU], xf[kU][TM];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int st = s0 + u * KW;
            const int64_t k0 = (int64_t)st * 128;
            const bool ok1 = st < nk && k0 + c1 < K, ok2 = st < nk && k0 + c2 < K;
            u32x4 lo = {0u, 0u, 0u, 0u}, hi = {0u, 0u, 0u, 0u};
            if (wrow_ok && ok1) lo = __builtin_nontemporal_load((const u32x4 *)(wrow + k0 + c1));
            if (wrow_ok && ok2) hi = __builtin_nontemporal_load((const u32x4 *)(wrow + k0 + c2));
            wf[u] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                const int64_t m = t * 16 + r;
                u32x4 xl = {0u, 0u, 0u, 0u}, xh = {0u, 0u, 0u, 0u};
                if (m < p.M && ok1) xl = *(const u32x4 *)(p.A + m * p.lda + k0 + c1);
                if (m < p.M && ok2) xh = *(const u32x4 *)(p.A + m * p.lda + k0 + c2);
                xf[u][t] = i32x8{(int)xl[0], (int)xl[1], (int)xl[2], (int)xl[3], (int)xh[0], (int)xh[1], (int)xh[2], (int)xh[3]};
            }
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            if (p.nan_zero) {
#pragma unroll
                for (int j = 0; j < 8; ++j) wf[u][j] = (int)scrub_nan4((uint32_t)wf[u][j]);
#pragma unroll
                for (int t = 0; t < TM; ++t)
#pragma unroll
                    for (int j = 0; j < 8; ++j) xf[u][t][j] = (int)scrub_nan4((uint32_t)xf[u][t][j]);
            }
#pragma unroll
            for (int t = 0; t < TM; ++t)
                acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[u], xf[u][t], acc[t], 0, 0, 0, kScaleOne, 0,
                                                                           kScaleOne);
        }
    }

    // K reduction across the waves
#pragma unroll
    for (int t = 0; t < TM; ++t) part[wave][t][lane] = acc[t];
    __syncthreads();
    // the KW waves finish the m-fragments t = kw, kw + KW, ...
    const bool has_bias = p.bias != nullptr, has_sr = p.scale_result != nullptr;
    const float sr = has_sr ? p.scale_result[0] : 1.0f;
    const int esz = p.out_dtype == FP8MI_F32 ? 4 : 2;
    const bool vec_ok = ((p.ldc * esz) % 16) == 0 && (((uintptr_t)p.C) % 16) == 0;
    for (int t = kw; t < TM; t += KW) {
        f32x4 s = part[0][t][lane];
#pragma unroll
        for (int w = 1; w < KW; ++w) s += part[w][t][lane];
        const int64_t m = t * 16 + r;
        if (m >= p.M) continue;
        const float sa = p.sa_row ? p.scale_a[m] : p.scale_a[0];
        const int64_t n = n0 + g * 4;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t nj = min(n + j, p.N - 1);
            const float sb = p.sb_row ? p.scale_b[nj] : p.scale_b[0];
            const float b = has_bias ? load_as_float(p.bias, p.transposed ? m : nj, p.bias_dtype) : 0.0f;
            v[j] = epilogue_value(s[j], sa, sb, has_bias, b, has_sr, sr, p.transposed != 0);
        }
        const int64_t idx = m * p.ldc + n;
        if (vec_ok && n + 3 < p.N && p.out_dtype == FP8MI_F32) {
            *(f32x4 *)((float *)p.C + idx) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (n + j < p.N) store_from_float(p.C, idx + j, v[j], p.out_dtype);
        }
    }
}

template <int TM, int kU, int KW>
int launch_kw(const MMParams &p, hipStream_t s)
{
    const int64_t grid = (p.N + 15) / 16;
    return fp8mi_launch(skinny_kernel<TM, kU, KW>, dim3((unsigned)grid), dim3(KW * 64), s, p);
}

template <int TM, int kU>
int launch(const MMParams &p, hipStream_t s)
{
    // as few waves per fragment as leaves each of them kU K-steps to keep in flight (K = 4096 with 8 waves left
    // each wave 4 of its 8 load slots); the grid stays one workgroup per 16 rows of W
    const int64_t nk = (p.K + 127) / 128;
    if (nk >= 8 * kU) return launch_kw<TM, kU, 8>(p, s);
    if (nk >= 4 * kU) return launch_kw<TM, kU, 4>(p, s);
    if (nk >= 2 * kU) return launch_kw<TM, kU, 2>(p, s);
    return launch_kw<TM, kU, 1>(p, s);
}

}  // namespace

bool fp8mi_skinny_supported(const MMParams &p)
{
    return p.M >= 1 && p.M <= 64 && p.K > 0 && (p.K % 16) == 0 && (p.lda % 16) == 0 && (p.ldb % 16) == 0 &&
           (((uintptr_t)p.A) & 15u) == 0 && (((uintptr_t)p.B) & 15u) == 0 && (p.N + 15) / 16 <= 0x7FFFFFFF;
}

int fp8mi_launch_skinny(const MMParams &p, hipStream_t s)
{
    if (p.M <= 16) return launch<1, 8>(p, s);   // 16 KiB of W in flight per wave
    if (p.M <= 32) return launch<2, 4>(p, s);
    if (p.M <= 48) return launch<3, 2>(p, s);
    return launch<4, 2>(p, s);
}
