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
//   * x is loaded FIRST (loads return in issue order: behind the W slab it would arrive last and push every multiply-add
//     into the kernel's tail) and handled once per workgroup, not once per row;
//   * the multiply-add has two forms.  fp32: x and W decoded with v_cvt_pk_f32_fp8 and accumulated with packed fp32 FMAs
//     (every e4m3 x e4m3 product is exact in fp32, so only the summation rounds) - 16 VALU instructions per KiB of W,
//     which showed as ~1 us of config C2 even when overlapped.  MFMA (K > 4096): the matrix core takes the raw bytes of a
//     lane as one operand and the x bytes of the same lane as the other; the DIAGONAL of the 16x16 product tile is the
//     lane-wise dot product, its trace the wave's - no decode or FMA instructions at all (see the kernel body);
//     accumulation is the fp8 matrix core's, as in the GEMM;
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

// OCC > 0: at most OCC workgroups of this kernel per CU (an LDS allocation of 160 KiB / OCC does it): bounds the bytes the CU has in flight
template <int STEPS, int RB, bool NT = true, int kWaves = 4, int ABL = 0, bool MFMA = false, int OCC = 0>
__global__ __launch_bounds__(kWaves * 64) void gemv_kernel(MMParams p_in)
{
    if constexpr (OCC > 0) {
        __shared__ int occ_pad[(160 * 1024 / OCC - 512) / 4];
        if (p_in.debug == 0x7FFFFFFF) occ_pad[threadIdx.x] = 1;   // (never true: keeps the allocation)
    }
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
    [[maybe_unused]] f32x4 macc[RB];  // MFMA form: 16x16 product tiles whose DIAGONALS are the wanted dot products (see below)
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        acc[r] = f32x2{0.0f, 0.0f};
        if constexpr (MFMA) macc[r] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }

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
        // 1. x for these k positions FIRST: loads return in issue order (vmcnt), so x behind the W slab would only arrive after
        //    the last W byte and every multiply-add would sit in the kernel's tail (measured: 2.5-3.5 us of 14.4 on config C2);
        //    x is 14 KiB from L2 and lands while the W loads are still being issued
        u32x4 xr[STEPS];
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            xr[i] = u32x4{0u, 0u, 0u, 0u};
            if (kv[i]) xr[i] = *(const u32x4 *)(x + kb[i]);
        }
        // 2. issue every W load of this chunk
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
        if (p.nan_zero) {  // the reference's NaN rule for x (fp8_matmul.metal:21), once per workgroup
#pragma unroll
            for (int i = 0; i < STEPS; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) xr[i][j] = scrub_nan4(xr[i][j]);
        }
        if constexpr (MFMA) {
            // 3m. the matrix core does the decode AND the multiply-add: v_mfma_scale_f32_16x16x128_f8f6f4 pairs byte j of
            //     lane (i, g) of its first operand with byte j of lane (i', g) of its second for every (i, i'); with the W bytes
            //     of a lane in the first operand and the x bytes of THE SAME lane (same k positions) in the second, the diagonal
            //     D[i][i] = sum over the four lanes i, i+16, i+32, i+48 of their 32-byte dot products - and the trace is the dot
            //     product over the 2 KiB the wave holds.  15/16 of the products are discarded, which the matrix pipe (4 % busy
            //     here) does not notice; what is gone is the VALU work: 16 instructions per KiB (8 v_cvt_pk_f32_fp8 + 8
            //     v_pk_fma_f32), measured 2.5-3.5 us of the 14.4 us of config C2 (timing-only build without the arithmetic:
            //     10.9-11.9 us, profiles/r02_gemv_shapes.txt).
#pragma unroll
            for (int r = 0; r < RB; ++r) {
#pragma unroll
                for (int i = 0; i < STEPS; i += 2) {
                    const u32x4 w1 = i + 1 < STEPS ? w[r][i + 1] : u32x4{0u, 0u, 0u, 0u};
                    const u32x4 x1 = i + 1 < STEPS ? xr[i + 1] : u32x4{0u, 0u, 0u, 0u};
                    const i32x8 a = {(int)w[r][i][0], (int)w[r][i][1], (int)w[r][i][2], (int)w[r][i][3], (int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
                    const i32x8 b = {(int)xr[i][0], (int)xr[i][1], (int)xr[i][2], (int)xr[i][3], (int)x1[0], (int)x1[1], (int)x1[2], (int)x1[3]};
                    macc[r] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, macc[r], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                }
            }
        } else {
            // 3. x decoded once for all RB rows (under the W stream)
            f32x2 xs[STEPS][8];
#pragma unroll
            for (int i = 0; i < STEPS; ++i) decode16(xr[i], xs[i]);
            // 4. consume in issue order
#pragma unroll
            for (int r = 0; r < RB; ++r) {
#pragma unroll
                for (int i = 0; i < STEPS; ++i) {
                    if constexpr (ABL == 1) {  // timing-only (diagnostic library): no decode, no FMA - the load structure alone
                        acc[r][0] += __uint_as_float((w[r][i][0] ^ w[r][i][1] ^ w[r][i][2] ^ w[r][i][3]) & 0x3FFFFFFFu);
                        continue;
                    }
                    f32x2 wf[8];
                    decode16(w[r][i], wf);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[r] = xs[i][j] * wf[j] + acc[r];
                }
            }
        }
    }

    if constexpr (MFMA) {
        // the diagonal of each 16x16 tile: element (row 4 (lane >> 4) + j, column lane & 15) is held by this lane's j-th
        // register, so the lanes with (lane & 15) >> 2 == lane >> 4 hold one diagonal element each (j = lane & 3)
        const int fr = lane & 15, fg = lane >> 4;
        const bool on_diag = (fr >> 2) == fg;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const float d = (fr & 2) ? ((fr & 1) ? macc[r][3] : macc[r][2]) : ((fr & 1) ? macc[r][1] : macc[r][0]);
            acc[r] = f32x2{on_diag ? d : 0.0f, 0.0f};
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

// ---------------------------------------------------------------------------------------------------------------
// 2 <= M <= 8 through the same weight-streaming structure ("few rows of x"): the reference sends these batches to its
// one-thread-per-output kernel (fp8_mps_native.py:208; batch 4 is one of its three published shapes).  W is streamed
// exactly as for M == 1 - 1-KiB wave-loads, x ahead of the slab, many light workgroups - and each pair of W wave-steps is
// multiplied with the matching bytes of EVERY x row by one MFMA per row (the diagonal trick above; the matrix pipe has the
// headroom: MX MFMAs per 2 KiB of W).  The skinny kernel (fragment-shaped loads, one workgroup per 16 rows of W) stays for
// 9 <= M <= 64.
// SWEEP: group g of workgroup b holds rows (g * gridDim + b) * RB ..: at any moment the workgroups in flight read ONE contiguous window of W that moves
// through the matrix (the access pattern of the fastest streaming-read probe, tools/probes/read_sweep.hip) instead of gridDim separate spans
template <int STEPS, int RB, int MX, int G, int kWaves = 4, bool SWEEP = false>
__global__ __launch_bounds__(kWaves * 64) void gemv_mx_kernel(MMParams p_in)
{
    // A workgroup owns G groups of RB consecutive rows of W and keeps its K-slices of all MX rows of x in registers across
    // them (x traffic from L2 = MX / (RB G) of the W stream; with RB G = 2 it was 4x the W bytes at M = 8 and the kernel ran
    // 3.5x slower than at M = 2).  The W loads of group g + 1 are issued before group g is multiplied.  Each lane's diagonal
    // elements go to LDS as they are (one masked ds_write per output cell and group instead of a 64-lane butterfly); the
    // sums over lanes and waves are formed once, after the last group.
    constexpr int kRows = RB * G;
    const MMParams p = pin_params(p_in);
    const float sr = p.scale_result ? p.scale_result[0] : 1.0f;
    const float sa0 = p.scale_a[0];
    const float sw0 = p.scale_b[0];
    constexpr int kThreads = kWaves * 64;
    __shared__ float part[kWaves][MX][kRows][16];
    __shared__ float dump[kWaves][64];
    __shared__ int dirty_cell[MX][kRows];

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * kRows;
    auto rown = [&](int r) -> int64_t {   // global row of the workgroup's r-th row (r = g * RB + row in group)
        return SWEEP ? ((int64_t)(r / RB) * gridDim.x + blockIdx.x) * RB + (r % RB) : row0 + r;
    };
    const int64_t K = p.K;
    const uint8_t *__restrict__ X = p.A;
    const uint8_t *__restrict__ W = p.B;
    const int M = (int)p.M;
    const int fr = lane & 15, fg = lane >> 4;
    const bool on_diag = (fr >> 2) == fg;   // this lane holds element (fr, fr) of every 16x16 tile in its register fr & 3

    f32x4 macc[MX][RB];
    {   // K <= kWaves * STEPS KiB (the launcher picks the shape): one pass, x stays in registers for all groups
        const int64_t kc = 0;
        int64_t kb[STEPS];
        bool kv[STEPS];
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            kb[i] = kc + (int64_t)(wave + kWaves * i) * 1024 + lane * 16;
            kv[i] = kb[i] < K;
        }
        u32x4 xr[MX][STEPS];   // x first (loads return in issue order), once per workgroup and chunk
#pragma unroll
        for (int m = 0; m < MX; ++m)
#pragma unroll
            for (int i = 0; i < STEPS; ++i) {
                xr[m][i] = u32x4{0u, 0u, 0u, 0u};
                if (m < M && kv[i]) xr[m][i] = *(const u32x4 *)(X + (int64_t)m * p.lda + kb[i]);
            }
        auto load_group = [&](int g, u32x4 (&w)[RB][STEPS]) {
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int64_t n = rown(g * RB + r);
                const uint8_t *wr = W + n * p.ldb;
#pragma unroll
                for (int i = 0; i < STEPS; ++i) {
                    if (n < p.N && kv[i]) w[r][i] = __builtin_nontemporal_load((const u32x4 *)(wr + kb[i]));
                    else w[r][i] = u32x4{0u, 0u, 0u, 0u};
                }
            }
        };
        auto mul_group = [&](int g, const u32x4 (&w)[RB][STEPS]) {
#pragma unroll
            for (int m = 0; m < MX; ++m)
#pragma unroll
                for (int r = 0; r < RB; ++r) macc[m][r] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < RB; ++r) {
#pragma unroll
                for (int i = 0; i < STEPS; i += 2) {
                    const u32x4 w1 = i + 1 < STEPS ? w[r][i + 1] : u32x4{0u, 0u, 0u, 0u};
                    const i32x8 a = {(int)w[r][i][0], (int)w[r][i][1], (int)w[r][i][2], (int)w[r][i][3], (int)w1[0], (int)w1[1], (int)w1[2], (int)w1[3]};
#pragma unroll
                    for (int m = 0; m < MX; ++m) {
                        const u32x4 x0 = xr[m][i], x1 = i + 1 < STEPS ? xr[m][i + 1] : u32x4{0u, 0u, 0u, 0u};
                        const i32x8 b = {(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], (int)x1[0], (int)x1[1], (int)x1[2], (int)x1[3]};
                        macc[m][r] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, macc[m][r], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
                    }
                }
            }
#pragma unroll
            for (int m = 0; m < MX; ++m)
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    const float d = (fr & 2) ? ((fr & 1) ? macc[m][r][3] : macc[m][r][2]) : ((fr & 1) ? macc[m][r][1] : macc[m][r][0]);
                    // NO divergent branch here: hipcc sinks the MFMAs (and the moves that zero their unused operand halves)
                    // into an `if (on_diag)` that holds their only use, the moves then run under a partial EXEC mask and
                    // the matrix core - which ignores EXEC - multiplies garbage in the other lanes.  Every lane stores;
                    // the lanes off the diagonal store to a dump slot.
                    float *cell = on_diag ? &part[wave][m][g * RB + r][fr] : &dump[wave][lane];
                    *cell = d;
                }
        };
        u32x4 wa[RB][STEPS], wb[RB][STEPS];
        load_group(0, wa);
        if (p.nan_zero) {   // the reference's NaN rule for x (fp8_matmul.metal:21), under the first W loads
#pragma unroll
            for (int m = 0; m < MX; ++m)
#pragma unroll
                for (int i = 0; i < STEPS; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) xr[m][i][j] = scrub_nan4(xr[m][i][j]);
        }
#pragma unroll
        for (int g = 0; g < G; g += 2) {
            if (g + 1 < G) load_group(g + 1, wb);
            mul_group(g, wa);
            if (g + 1 < G) {
                if (g + 2 < G) load_group(g + 2, wa);
                mul_group(g + 1, wb);
            }
        }
    }
    __syncthreads();

    int dirty = 0;
    // kP adjacent lanes per output cell: each adds its share of the cell's kWaves x 16 lane sums, a DPP group sum finishes (round 3: one thread per
    // cell added 64-128 LDS values in series in the tail of every workgroup)
    constexpr int kCells = MX * kRows, kVals = kWaves * 16;
    constexpr int kP = kThreads / kCells >= 16 ? 16 : (kThreads / kCells >= 8 ? 8 : (kThreads / kCells >= 4 ? 4 : (kThreads / kCells >= 2 ? 2 : 1)));
    const int cell = threadIdx.x / kP, cj = threadIdx.x % kP;
    const int cm = cell / kRows, cr = cell % kRows;  // this thread's output cell
    auto finish = [&](int m, int r, float sum) {
        const int64_t n = rown(r);
        const float sa = p.sa_row ? p.scale_a[m] : sa0;
        const float sw = p.sb_row ? p.scale_b[n] : sw0;
        const float b = p.bias ? load_as_float(p.bias, p.transposed ? (int64_t)m : n, p.bias_dtype) : 0.0f;
        store_from_float(p.C, (int64_t)m * p.ldc + n,
                         epilogue_value(sum, sa, sw, p.bias != nullptr, b, p.scale_result != nullptr, sr, p.transposed != 0), p.out_dtype);
    };
    static_assert(MX * kRows <= kThreads, "one thread per output cell of the workgroup");
    if (threadIdx.x < kCells * kP) {   // (whole groups of kP lanes: kCells * kP is a multiple of kP)
        float sum = 0.0f;
#pragma unroll
        for (int v = 0; v < kVals / kP; ++v) {
            const int i = v * kP + cj;
            sum += part[i >> 4][cm][cr][i & 15];
        }
        sum = group_sum<kP>(sum);
        dirty = (p.nan_zero && sum != sum) ? 1 : 0;
        if (cj == 0) {
            dirty_cell[cm][cr] = dirty;
            if (cm < M && rown(cr) < p.N && !dirty) finish(cm, cr, sum);
        }
    }
    if (!__syncthreads_or(dirty)) return;

    // rare path: rows of W holding NaN bytes, byte-wise reference decode
    float *red = &part[0][0][0][0];
    for (int m = 0; m < M; ++m)
        for (int r = 0; r < kRows; ++r) {
            if (!dirty_cell[m][r]) continue;  // block-uniform
            const int64_t n = rown(r);
            if (n >= p.N) continue;
            const uint8_t *wr = W + n * p.ldb;
            const uint8_t *xm = X + (int64_t)m * p.lda;
            float sv = 0.0f;
            for (int64_t k = threadIdx.x; k < K; k += kThreads) sv += decode_ref(xm[k]) * decode_ref(wr[k]);
            sv = wave_sum(sv);
            __syncthreads();
            if (lane == 0) red[wave] = sv;
            __syncthreads();
            if (threadIdx.x == 0) {
                float sum = 0.0f;
#pragma unroll
                for (int wv = 0; wv < kWaves; ++wv) sum += red[wv];
                finish(m, r, sum);
            }
        }
}

template <int STEPS, int RB, int MX, int G, int kWaves = 4, bool SWEEP = false>
int launch_mx(const MMParams &p, hipStream_t s)
{
    const int64_t grid = (p.N + RB * G - 1) / (RB * G);
    return fp8mi_launch(gemv_mx_kernel<STEPS, RB, MX, G, kWaves, SWEEP>, dim3((unsigned)grid), dim3(kWaves * 64), s, p);
}

template <int STEPS, int RB, bool NT = true, int kWaves = 4, int ABL = 0, bool MFMA = false, int OCC = 0>
int launch(const MMParams &p, hipStream_t s)
{
    const int64_t grid = (p.N + RB - 1) / RB;
    return fp8mi_launch(gemv_kernel<STEPS, RB, NT, kWaves, ABL, MFMA, OCC>, dim3((unsigned)grid), dim3(kWaves * 64), s, p);
}

}  // namespace

bool fp8mi_gemv_supported(const MMParams &p)
{
    return p.M == 1 && p.K > 0 && (p.K % 16) == 0 && (p.ldb % 16) == 0 && (((uintptr_t)p.A) & 15u) == 0 &&
           (((uintptr_t)p.B) & 15u) == 0 && (p.N + 3) / 4 <= 0x7FFFFFFF;
}

bool fp8mi_gemv_mx_supported(const MMParams &p)
{
    return p.M >= 2 && p.M <= 8 && p.K > 0 && p.K <= 16384 && (p.K % 16) == 0 && (p.lda % 16) == 0 && (p.ldb % 16) == 0 && (((uintptr_t)p.A) & 15u) == 0 &&
           (((uintptr_t)p.B) & 15u) == 0 && (p.N + 1) / 2 <= 0x7FFFFFFF;
}

int fp8mi_launch_gemv_mx_variant(const MMParams &p, int id, hipStream_t s);

int fp8mi_launch_gemv_mx(const MMParams &p, hipStream_t s)
{
    const int64_t steps = (p.K + 4095) / 4096;   // wave-steps per wave with 4 waves; K <= 16384 (fp8mi_gemv_mx_supported)
    if (p.M <= 2) {
        if (steps <= 1) return launch_mx<1, 2, 2, 4>(p, s);
        if (steps <= 2) return launch_mx<2, 2, 2, 4>(p, s);
        return launch_mx<4, 2, 2, 4>(p, s);
    }
    if (p.M <= 4) {
        if (steps <= 1) return launch_mx<1, 2, 4, 4>(p, s);
        if (steps <= 2) return launch_mx<2, 2, 4, 4>(p, s);
        return launch_mx<4, 2, 4, 4>(p, s);
    }
    if (steps <= 1) return launch_mx<1, 2, 8, 4>(p, s);
    if (steps <= 2) return launch_mx<2, 2, 8, 8>(p, s);
    return launch_mx<2, 2, 8, 8, 8>(p, s);   // 8 x rows, deep K: 8 waves x two wave-steps (x alone is 64 registers per lane)
}

#ifdef FP8MI_DIAG
// few-rows kernel: <steps, rows per group, x rows, groups per workgroup>; id = 70 + variant, the x-row count follows M
int fp8mi_launch_gemv_mx_variant(const MMParams &p, int id, hipStream_t s)
{
    const int64_t steps = (p.K + 4095) / 4096;
    const bool sweep = id >= 74;   // 74..77: the same with the groups of a workgroup gridDim apart (one moving read window)
    const int v = sweep ? id - 74 : id - 70;
    const int g = v == 0 ? 1 : (v == 1 ? 2 : (v == 2 ? 4 : 8));
#define MXV(S, MXN)                                                             \
    (sweep ? (g == 1 ? launch_mx<S, 2, MXN, 1, 4, true>(p, s) : g == 2 ? launch_mx<S, 2, MXN, 2, 4, true>(p, s) \
              : g == 4 ? launch_mx<S, 2, MXN, 4, 4, true>(p, s) : launch_mx<S, 2, MXN, 8, 4, true>(p, s))        \
           : (g == 1 ? launch_mx<S, 2, MXN, 1>(p, s) : g == 2 ? launch_mx<S, 2, MXN, 2>(p, s) \
              : g == 4 ? launch_mx<S, 2, MXN, 4>(p, s) : launch_mx<S, 2, MXN, 8>(p, s)))
    if (p.M <= 2) return steps <= 1 ? MXV(1, 2) : steps <= 2 ? MXV(2, 2) : MXV(4, 2);
    if (p.M <= 4) return steps <= 1 ? MXV(1, 4) : steps <= 2 ? MXV(2, 4) : MXV(4, 4);
    if (steps <= 2) return steps <= 1 ? MXV(1, 8) : MXV(2, 8);
    return g == 1 ? launch_mx<2, 2, 8, 1, 8>(p, s) : g == 2 ? launch_mx<2, 2, 8, 2, 8>(p, s) : g == 4 ? launch_mx<2, 2, 8, 4, 8>(p, s) : launch_mx<2, 2, 8, 8, 8>(p, s);
#undef MXV
}
#endif

#ifdef FP8MI_DIAG  // launch-shape variants for A/B timing (diagnostic library only): <K-steps per wave, rows per workgroup, nt, waves>
int fp8mi_launch_gemv_variant(const MMParams &p, int id, hipStream_t s)
{
    switch (id) {
    case 40: return launch<4, 8>(p, s);               // the product's shape for C2
    case 41: return launch<4, 4>(p, s);
    case 42: return launch<2, 8, true, 8>(p, s);
    case 43: return launch<2, 4, true, 8>(p, s);
    case 44: return launch<4, 8, false>(p, s);        // default cache policy instead of nt
    case 45: return launch<2, 2, true, 8>(p, s);
    case 46: return launch<4, 2>(p, s);
    case 47: return launch<1, 8, true, 16>(p, s);     // 16 waves x 1 KiB, 8 rows
    case 48: return launch<2, 4, true, 8, 1>(p, s);   // <2,4,8 waves> with the arithmetic removed (timing only)
    case 49: return launch<4, 8, true, 4, 1>(p, s);   // the product's shape with the arithmetic removed (timing only)
    case 50: return launch<4, 8, true, 4, 0, true>(p, s);   // MFMA accumulate, <4,8>
    case 51: return launch<2, 4, true, 8, 0, true>(p, s);   // MFMA accumulate, 8 waves x 2 steps, 4 rows
    case 52: return launch<4, 4, true, 4, 0, true>(p, s);   // MFMA accumulate, <4,4>
    case 53: return launch<2, 8, true, 8, 0, true>(p, s);   // MFMA accumulate, 8 waves x 2 steps, 8 rows
    case 54: return launch<4, 2, true, 4, 0, true>(p, s);
    case 55: return launch<2, 2, true, 8, 0, true>(p, s);
    case 56: return launch<2, 4, true, 4, 0, true>(p, s);   // K <= 8192
    case 57: return launch<2, 8, true, 4, 0, true>(p, s);
    case 58: return launch<1, 4, true, 4, 0, true>(p, s);   // K <= 4096
    case 59: return launch<1, 2, true, 4, 0, true>(p, s);
    case 60: return launch<1, 8, true, 4, 0, true>(p, s);
    case 61: return launch<1, 2>(p, s);
    case 180: return launch<1, 1>(p, s);
    case 181: return launch<1, 8>(p, s);
    case 182: return launch<1, 2, true, 4, 0, false, 4>(p, s);   // <1,2> with at most 4 workgroups per CU
    case 183: return launch<1, 4, false>(p, s);                  // default cache policy
    case 62: return launch<2, 2, true, 4, 0, true>(p, s);
    case 63: return launch<4, 1, true, 4, 0, true>(p, s);
    case 64: return launch<1, 1, true, 4, 0, true>(p, s);
    case 65: return launch<4, 2, true, 4, 0, true, 2>(p, s);   // <4,2> MFMA with at most 2 / 3 / 4 / 6 workgroups per CU
    case 66: return launch<4, 2, true, 4, 0, true, 3>(p, s);
    case 67: return launch<4, 2, true, 4, 0, true, 4>(p, s);
    case 68: return launch<4, 2, true, 4, 0, true, 6>(p, s);
    case 69: return launch<4, 1, true, 4, 0, true, 4>(p, s);   // <4,1> with at most 4 / 6 per CU
    case 160: return launch<4, 1, true, 4, 0, true, 6>(p, s);
    case 161: return launch<4, 4, true, 4, 0, true, 2>(p, s);
    // M == 1 on the few-rows kernel's structure (x kept in registers across G groups of 2 rows, next group's loads under this group's MFMAs):
    // x traffic from L2 = 1 / (2 G) of the W stream instead of 1 / 2
    case 162: return launch_mx<4, 2, 1, 2>(p, s);
    case 163: return launch_mx<4, 2, 1, 4>(p, s);
    case 164: return launch_mx<4, 2, 1, 8>(p, s);
    case 165: return launch_mx<4, 1, 1, 8>(p, s);
    case 166: return launch_mx<4, 1, 1, 4>(p, s);
    case 167: return launch_mx<4, 1, 1, 16>(p, s);
    case 168: return launch_mx<2, 1, 1, 8>(p, s);     // K <= 8192
    case 169: return launch_mx<2, 2, 1, 4>(p, s);
    case 170: return launch_mx<2, 1, 1, 8, 8>(p, s);  // 8 waves x 2 steps: K <= 16384
    case 171: return launch_mx<2, 2, 1, 4, 8>(p, s);
    case 174: return launch_mx<4, 1, 1, 4, 4, true>(p, s);   // ... with the rows of a workgroup dealt out gridDim apart (one moving window)
    case 175: return launch_mx<4, 1, 1, 8, 4, true>(p, s);
    case 176: return launch_mx<4, 2, 1, 4, 4, true>(p, s);
    case 177: return launch_mx<4, 2, 1, 2, 4, true>(p, s);
    case 178: return launch_mx<4, 1, 1, 16, 4, true>(p, s);
    case 179: return launch_mx<2, 2, 1, 4, 4, true>(p, s);   // K <= 8192
    case 172: return launch_mx<1, 2, 1, 4>(p, s);     // K <= 4096
    case 173: return launch_mx<1, 1, 1, 8>(p, s);
    default: return FP8MI_E_ENUM;
    }
}
#endif

int fp8mi_launch_gemv(const MMParams &p, bool fp32_only, hipStream_t s)
{
    // wave-steps per wave for one pass over K (4 waves x 1 KiB per step)
    const int64_t steps = (p.K + 4095) / 4096;
    // Launch shape and arithmetic, measured on MI355X with x loaded ahead of the W slab (tools/ab_kernels.py on the
    // diagnostic library, profiles/r02_gemv_shapes.txt).  Few rows per workgroup = many light workgroups, which is what a
    // 5-30 us streaming kernel wants; past one wave-step per wave the matrix core takes over the multiply-add:
    //   K <= 4096:  fp32 FMA, 4 rows   (K = N = 4096: 5.2-5.5 us for either form; N = 14336: 11.0 vs 12.5 us MFMA)
    //   K <= 8192:  MFMA, 2 rows       (K = N = 8192: 11.5 vs 12.9 us fp32)
    //   K  > 8192:  MFMA, 2 rows       (C2 K = 14336, N = 4096: 10.95 vs 12.0 us fp32 <4,2>, 13.2 us fp32 <4,8>; N = 14336: 30.7 vs 34.1 us)
    // fp32_only (FP8MI_KERNEL_GEMV_FP32) keeps IEEE fp32 accumulation at every K.
    if (steps <= 1) return launch<1, 4>(p, s);
    if (fp32_only) {
        if (steps <= 2) return launch<2, 4>(p, s);
        if (p.K > 16384) return launch<2, 4, true, 8>(p, s);  // 8 waves x 2 steps, loops over 16-KiB chunks
        return launch<4, 2>(p, s);
    }
    if (steps <= 2) return launch<2, 2, true, 4, 0, true>(p, s);
    // Round 3: deep K against a matrix of few rows (config C2) on the few-rows kernel's structure with ONE row of x - x stays in registers
    // across 4 groups of 2 weight rows (x traffic from L2: 1/8 of the W stream instead of 1/2), the next group's loads go out under this
    // group's MFMAs, and the groups of a workgroup are gridDim apart, so that the workgroups in flight read one contiguous window that moves
    // through W (the walk of the fastest read kernel of tools/probes/read_sweep.hip).  C2: 10.82 against 11.12 us interleaved; K = N = 14336
    // is 3-4 % slower that way (31.5 against 30.2) and stays on the plain form.
    if (p.K <= 16384 && p.N <= 8192 && p.lda % 16 == 0) return launch_mx<4, 2, 1, 4, 4, true>(p, s);
    return launch<4, 2, true, 4, 0, true>(p, s);
}
