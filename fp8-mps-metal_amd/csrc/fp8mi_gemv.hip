// M == 1 FP8 vec-mat for gfx950:  out[n] = (sum_k dec(x[k]) dec(W[n,k])) * sx * sw[n]
//
// Replaces fp8_scaled_vecmat_kernel (fp8_matmul.metal:155-210: one 32-lane
// simdgroup per output row, 4 scalar byte loads per lane per step, software
// decode of both operands).  Here the kernel is designed around HBM3E:
//
//   * a workgroup of 4 (8 for K > 16384) wave64s owns RB consecutive rows of W; the waves
//     split K between them in 1-KiB wave-steps (lane l of a wave reads the 16
//     bytes at k = step*1024 + 16 l: one dwordx4 per lane, a full contiguous
//     KiB per wave-instruction, non-temporal because W is streamed once);
//   * every load of the workgroup's W slab is issued before the first use
//     (RB x STEPS dwordx4 per lane, compile-time unrolled), so the whole
//     matrix is in flight within the first microsecond of the launch - at
//     M == 1 the kernel lasts ~10 us and is latency-, not issue-, bound;
//   * x is decoded ONCE per workgroup into registers (v_cvt_pk_f32_fp8), not
//     once per row; W bytes are decoded with the same instruction and
//     accumulated with packed fp32 FMAs (every e4m3 x e4m3 product is exact in
//     fp32, so only the summation rounds);
//   * K is reduced across lanes by a wave64 butterfly, across the four waves
//     through LDS; the scale / bias / result-scale / cast epilogue is fused.
//
// NaN bytes (0x7F/0xFF) must decode to 0.0 (fp8_matmul.metal:21) while the
// hardware convert yields NaN.  x is scrubbed while it is decoded (once).  W is
// not scrubbed in the streaming loop: a NaN byte in row n makes out[n] NaN
// (finite e4m3 products cannot overflow fp32), which is detected after the
// reduction; only such rows are recomputed with the exact byte-wise decode.

#include "fp8mi_common.h"

namespace {


FP8MI_DEVICE void decode16(const u32x4 &w, f32x2 (&f)[8])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = __builtin_amdgcn_cvt_pk_f32_fp8(w[j], false);
        f[2 * j + 1] = __builtin_amdgcn_cvt_pk_f32_fp8(w[j], true);
    }
}

template <int STEPS, int RB, bool NT = true, int kWaves = 4>
__global__ __launch_bounds__(kWaves * 64) void gemv_kernel(MMParams p_in)
{
    const MMParams p = pin_params(p_in);  // every kernel argument in one scalar-load clause (fp8mi_common.h)
    // epilogue scalars: fetched now, under the weight stream (loaded where they are used they were a dependent global load
    // between the reduction and the store)
    const float sr = p.scale_result ? p.scale_result[0] : 1.0f;
    const float sx = p.scale_a[0];
    const float sw0 = p.scale_b[0];
    constexpr int kThreads = kWaves * 64;
    __shared__ float part[kWaves][RB];
    __shared__ int dirty_rows[RB];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * RB;
    const int64_t K = p.K;
    const uint8_t *__restrict__ x = p.A;
    const uint8_t *__restrict__ W = p.B;

    f32x2 acc[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) acc[r] = f32x2{0.0f, 0.0f};

    // chunk of K handled per outer iteration: 4 waves x STEPS wave-steps of 1 KiB
    constexpr int64_t kChunk = (int64_t)kWaves * STEPS * 1024;
    for (int64_t kc = 0; kc < K; kc += kChunk) {
        int64_t kb[STEPS];
        bool kv[STEPS];
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            kb[i] = kc + (int64_t)(wave + kWaves * i) * 1024 + lane * 16;
            kv[i] = kb[i] < K;  // K % 16 == 0: a 16-byte piece is all in or all out
        }
        // 1. issue every W load of this chunk
        u32x4 w[RB][STEPS];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const bool rv = row0 + r < p.N;
            const uint8_t *wr = W + (row0 + r) * p.ldb;
#pragma unroll
            for (int i = 0; i < STEPS; ++i) {
                if (rv && kv[i]) w[r][i] = NT ? __builtin_nontemporal_load((const u32x4 *)(wr + kb[i])) : *(const u32x4 *)(wr + kb[i]);
                else w[r][i] = u32x4{0u, 0u, 0u, 0u};
            }
        }
        // 2. x for the same k positions, decoded once for all RB rows
        f32x2 xs[STEPS][8];
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            u32x4 xv = u32x4{0u, 0u, 0u, 0u};
            if (kv[i]) xv = *(const u32x4 *)(x + kb[i]);
            if (p.nan_zero) {
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[j] = scrub_nan4(xv[j]);
            }
            decode16(xv, xs[i]);
        }
        // 3. consume in issue order
#pragma unroll
        for (int r = 0; r < RB; ++r) {
#pragma unroll
            for (int i = 0; i < STEPS; ++i) {
                f32x2 wf[8];
                decode16(w[r][i], wf);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[r] = xs[i][j] * wf[j] + acc[r];
            }
        }
    }

    // K reduction: lanes, then waves
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        float s = wave_sum(acc[r][0] + acc[r][1]);
        if (lane == 0) part[wave][r] = s;
    }
    __syncthreads();

    int dirty = 0;
    if (threadIdx.x < RB) {
        const int r = threadIdx.x;
        const int64_t n = row0 + r;
        float sum = 0.0f;
#pragma unroll
        for (int wv = 0; wv < kWaves; ++wv) sum += part[wv][r];
        dirty = (p.nan_zero && sum != sum) ? 1 : 0;
        dirty_rows[r] = dirty;
        if (n < p.N && !dirty) {
            const float sw = p.sb_row ? p.scale_b[n] : sw0;
            const float b = p.bias ? load_as_float(p.bias, p.transposed ? 0 : n, p.bias_dtype) : 0.0f;
            store_from_float(p.C, n, epilogue_value(sum, sx, sw, p.bias != nullptr, b, p.scale_result != nullptr, sr, p.transposed != 0),
                             p.out_dtype);
        }
    }
    if (!__syncthreads_or(dirty)) return;

    // rare path: rows of W holding NaN bytes, byte-wise reference decode
    for (int r = 0; r < RB; ++r) {
        if (!dirty_rows[r]) continue;  // block-uniform
        const int64_t n = row0 + r;
        const uint8_t *wr = W + n * p.ldb;
        float s = 0.0f;
        for (int64_t k = threadIdx.x; k < K; k += kThreads) s += decode_ref(x[k]) * decode_ref(wr[k]);
        s = wave_sum(s);
        __syncthreads();
        if (lane == 0) part[wave][0] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            float sum = 0.0f;
#pragma unroll
            for (int wv = 0; wv < kWaves; ++wv) sum += part[wv][0];
            const float sw = p.sb_row ? p.scale_b[n] : sw0;
            const float b = p.bias ? load_as_float(p.bias, p.transposed ? 0 : n, p.bias_dtype) : 0.0f;
            store_from_float(p.C, n, epilogue_value(sum, sx, sw, p.bias != nullptr, b, p.scale_result != nullptr, sr, p.transposed != 0),
                             p.out_dtype);
        }
    }
}

template <int STEPS, int RB, bool NT = true, int kWaves = 4>
int launch(const MMParams &p, hipStream_t s)
{
    const int64_t grid = (p.N + RB - 1) / RB;
    return fp8mi_launch(gemv_kernel<STEPS, RB, NT, kWaves>, dim3((unsigned)grid), dim3(kWaves * 64), s, p);
}

}  // namespace

bool fp8mi_gemv_supported(const MMParams &p)
{
    return p.M == 1 && p.K > 0 && (p.K % 16) == 0 && (p.ldb % 16) == 0 && (((uintptr_t)p.A) & 15u) == 0 &&
           (((uintptr_t)p.B) & 15u) == 0 && (p.N + 3) / 4 <= 0x7FFFFFFF;
}

int fp8mi_launch_gemv(const MMParams &p, hipStream_t s)
{
    // wave-steps per wave for one pass over K (4 waves x 1 KiB per step)
    const int64_t steps = (p.K + 4095) / 4096;
    // rows per workgroup: few rows = many small workgroups, which is what a 5-15 us kernel wants (measured with
    // tools/time_shape.py; RB = 16 -> 4 at K = 4096: N = 14336 16.3 -> 12.2 us, N = 4096 8.4 -> 5.9 us; K = 8192: 15.8
    // -> 15.0 us); with four wave-steps per wave (K > 8192) 8 rows are better until the grid gets short of ~450
    // workgroups (K = 14336, N = 4096: 14.5 vs 15.5 us; K = 12288, N = 3072: 11.8 vs 10.4 us)
    if (steps <= 1) return launch<1, 4>(p, s);
    if (steps <= 2) return launch<2, 4>(p, s);
    if ((p.N + 7) / 8 < (7 * (int64_t)fp8mi_cu_count()) / 4) return launch<4, 4>(p, s);
    if (p.K > 16384) return launch<2, 4, true, 8>(p, s);  // 8 waves x 2 steps, loops over 16-KiB chunks (K = 28672, N = 8192: 41 vs 44.6 us)
    return launch<4, 8>(p, s);
}
