// 256x256-tile FP8 e4m3fn GEMM with ONE wave per SIMD and a hand-scheduled K loop (gfx950).
//
// Same contract and the same bits as the ring kernels of fp8mi_gemm.hip (same LDS image, same fragment -> MFMA operand
// map, K-steps added in order), for shapes of whole 128-byte K-steps (any M, N):
//
//     C[m,n] = cast(((sum_k dec(A[m,k]) dec(B[n,k])) * sa[m] * sb[n] + bias[n]) * sr)        (fp8_matmul.metal:99-147)
//
// Why a second kernel: with 8 waves of 128x64 (fp8mi_gemm.hip) every K-step moves 192 KiB of fragments out of the LDS
// and the two waves of a SIMD serialise on barriers - the matrix pipe is busy 51 % of the time on the FLUX shape
// (profiles/r02_pmc_gemm_flux.txt).  Here each of 4 waves owns 128x128 outputs: 256 accumulators in AGPRs and 128
// fragment VGPRs - the SIMD's whole 512-entry register file - 128 KiB of fragment reads per step, and the reads of step
// t+1, the global->LDS stream of step t+2 and the 64 MFMAs of step t are interleaved in ONE instruction stream.  hipcc
// cannot be made to emit that stream, so the loop is generated assembly with fixed registers:
// csrc/gen/gen_gemm256_loop.py -> fp8mi_gemm256_loop.inc.  Everything around the loop is ordinary HIP.
//
// NaN bytes (reference: decode to 0.0, fp8_matmul.metal:21): as in the ring kernels the loop runs unscrubbed; a NaN
// accumulator proves a NaN byte took part and only then the workgroup redoes its tile with the scrubbing loop.

#include "fp8mi_gemm_epi.h"
#include "fp8mi_gemm256_loop.inc"

// The generated statements list m0 (written by their LDS-DMA groups) as a clobber so that LLVM sees a definition of M0 there and never
// carries an M0 value of its own across them; clang remarks that m0 is a reserved register (it will not be saved / restored - none is needed)
#pragma clang diagnostic ignored "-Winline-asm"

namespace {

constexpr int kBM = 256, kThreads256 = 256;   // 4 waves, one per SIMD
typedef float f32x32 __attribute__((ext_vector_type(32)));   // 32 accumulators (asm operand type)
// Two workgroup tiles share everything but the wave tile's width: BN = 256 (wave 128x128, 64 MFMAs per K-step: the matrix-bound
// shape) and BN = 128 (wave 128x64, 32 MFMAs per step, 48-KiB stages: for shapes that give 256x256 tiles less than a round).
template <int BN>
struct Geo {
    static constexpr int kSlotBytes = (kBM + BN) * BK;   // one K-step: A's 256 rows, then B's
    static constexpr int kRing = 2 * kSlotBytes;
    static constexpr int kCols = BN / 2;                 // columns of a wave tile
    static constexpr int kDumpRow = 16 * kCols * 4;      // one fragment row of the accumulator dump (16 rows of fp32)
    static constexpr int kDumpWave = 4 * kDumpRow;       // a wave's share of the ring: 4 fragment rows
    static constexpr int kTabBase = kRing + kFlagBytes;
};

// The epilogue's rounding sequence is the ring kernels': (acc * sa) * sb, + bias, * sr - four roundings.  With the
// switches as template parameters hipcc would contract the multiply-add into an fma (one rounding less, different bits).
#pragma clang fp contract(off)

// Per-wave epilogue tables in the LDS behind the ring, fp32, one entry per row / column of the wave tile: scale_a (or the
// per-tensor value), scale_b, bias along n, bias along m (transposed).  Each lane fetches its two entries of every table
// at kernel entry (loads in flight under the K loop, uniform type switches out of the epilogue's inner loop) and writes
// them to the LDS after the loop.
constexpr int kTabBytes = 4 * 512 + 16;                  // per wave: four 128-entry tables, then scale_result
// the launch's uniform switches packed into one SGPR (ten separate fields kept alive across the tile loop ran hipcc out of SGPRs:
// it parked booleans in VGPRs, spilled those to scratch and reloaded them - with a full vmcnt wait - in front of the K loop)
enum { kFBias = 1, kFTransposed = 2, kFSaRow = 4, kFSbRow = 8, kFNanZero = 16, kFSr = 32, kFOutShift = 6, kFBiasTypeShift = 8 };

struct TabRegs {       // as loaded: nothing here is USED before the K loop (a use would wait for the load in front of it)
    float sa[2], sb[2], sr;
    uint32_t bias[2];  // raw fp32 bits, or a zero-extended 16-bit pattern
};

FP8MI_DEVICE TabRegs load_tables(const MMParams &p, int flags, int64_t m_wave, int64_t n_wave, int cols, int lane)
{
    TabRegs t;
    // rows of a ragged last m-tile beyond M take the entry of row M - 1 (any valid address: they are never stored); the clamp is
    // formed from scalars and one 32-bit min per lane (as 64-bit vector arithmetic its constants were hoisted and spilled)
    const int m_last = (int)p.M - 1, m_base = min((int)m_wave, m_last);   // (32-bit: M < 2^31, fp8mi_gemm256_supported)
    const int m_lim = max(min(m_last - m_base, 127), 0);
    const int n_last = (int)p.N - 1, n_base = min((int)n_wave, n_last), n_lim = max(min(n_last - n_base, cols - 1), 0);   // the same for columns
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int i = lane + 64 * h;
        t.sa[h] = p.scale_a[(flags & kFSaRow) ? m_base + min(i, m_lim) : 0];
        t.sb[h] = p.scale_b[(flags & kFSbRow) ? n_base + min(i, n_lim) : 0];
        t.bias[h] = 0u;
    }
    t.sr = 1.0f;
    if (flags & kFSr) t.sr = p.scale_result[0];
    if (flags & kFBias) {
        const int64_t i0 = (flags & kFTransposed) ? m_base + min(lane, m_lim) : n_base + min(lane, n_lim);
        const int64_t i1 = (flags & kFTransposed) ? m_base + min(lane + 64, m_lim) : n_base + min(lane + 64, n_lim);
        if (((flags >> kFBiasTypeShift) & 3) == FP8MI_F32) {
            t.bias[0] = ((const uint32_t *)p.bias)[i0];
            t.bias[1] = ((const uint32_t *)p.bias)[i1];
        } else {
            t.bias[0] = ((const uint16_t *)p.bias)[i0];
            t.bias[1] = ((const uint16_t *)p.bias)[i1];
        }
    }
    return t;
}

FP8MI_DEVICE void store_tables(int flags, const TabRegs &t, float *tab, int lane)
{
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float b = 0.0f;
        const int bt = (flags >> kFBiasTypeShift) & 3;
        if (flags & kFBias) {
            if (bt == FP8MI_F32) b = __builtin_bit_cast(float, t.bias[h]);
            else if (bt == FP8MI_BF16) b = __builtin_bit_cast(float, t.bias[h] << 16);
            else b = (float)__builtin_bit_cast(_Float16, (uint16_t)t.bias[h]);
        }
        tab[lane + 64 * h] = t.sa[h];
        tab[128 + lane + 64 * h] = t.sb[h];
        tab[256 + lane + 64 * h] = (flags & kFTransposed) ? 0.0f : b;
        tab[384 + lane + 64 * h] = (flags & kFTransposed) ? b : 0.0f;
    }
    tab[512] = t.sr;
}

// Fused epilogue of one HALF of the wave tile (fragment rows 4 half .. 4 half + 3, i.e. 64 rows x 128 columns), read back
// ROW-WISE from the accumulator dump (gen_gemm256_loop.py).  A lane takes the 32 bytes at pair position pp of row r: the
// chunks at positions 2 pp and 2 pp + 1, which hold the column chunks 2 (pp ^ (r >> 1)) + (r & 1) and its neighbour - eight
// consecutive columns after an exchange in odd rows - so every global store instruction writes whole rows (16 lanes x
// 32 B of fp32 or x 16 B of 16-bit types) without a second trip through the LDS.  Four rows per instruction, four
// instructions' LDS reads in flight at a time (one wave per SIMD: nothing else hides their latency).  TABLES = false
// (per-tensor scales): the scale tables are not read per element.  Returns the sum of everything read (NaN vote).
typedef __attribute__((address_space(3))) const uint8_t lds_cu8;
typedef __attribute__((address_space(3))) const float lds_cf32;
typedef __attribute__((address_space(3))) const f32x4 lds_cf32x4;
typedef __attribute__((address_space(1))) uint8_t glb_u8;

template <int COLS, int OUT, bool BIAS, bool TRANSPOSED, bool TABLES>
FP8MI_DEVICE f32x4 epilogue_half(const MMParams &p, int flags, float sr, lds_cu8 *dump, lds_cf32 *tab, int half, int64_t m_wave, int64_t n_wave, int64_t ldc, int rows_ok, int cols_ok, int lane)
{
    constexpr int kEsz = OUT == FP8MI_F32 ? 4 : 2;
    constexpr int kBatch = (BIAS && !TABLES) ? 2 : 4;   // store instructions per batch (with a bias two batches of four in flight do not fit the registers the asm leaves)
    constexpr int kLpr = COLS / 8, kRpi = 64 / kLpr, kIters = 64 / kRpi;   // lanes per row, rows per instruction, instructions per half
    constexpr int kDumpRow = 16 * COLS * 4;
    const int pp = lane % kLpr, rsub = lane / kLpr;
    // fp32 output (round 3): a lane's 32 bytes would go out as two 16-byte stores that each leave 16-byte holes in every line they touch
    // (streamed through the L2 on a 256 MiB C that cost 14 %: M = N = K = 8192 525 us against 450 with write-back stores).  Instead a lane
    // takes ONE 16-byte chunk (position P) from each of two rows kRpiF apart, so that each store instruction writes kRpiF whole rows.
    constexpr bool kF32 = OUT == FP8MI_F32;
    constexpr int kLprF = COLS / 4, kRpiF = 64 / kLprF;
    const int P = lane % kLprF, rsubF = lane / kLprF;
    auto chunk_col = [&](int r) { return (2 * (((P >> 1) ^ (r >> 1)) & (kLpr - 1)) + ((P ^ r) & 1)) * 4; };   // first column of the chunk at position P of row r
    // the wave tile of C as a raw buffer: 32-bit offsets in the store instead of 64-bit pointer arithmetic per row
    // (num_records = the wave tile's valid rows: stores to rows of a ragged last m-tile beyond M are out of range and dropped)
    __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)((uint8_t *)p.C + (m_wave * ldc + n_wave) * kEsz), 0,
                                                                  (int)((int64_t)rows_ok * ldc * kEsz), 0x00020000);
#ifndef FP8MI_EPI_AUX
#define FP8MI_EPI_AUX kCStoreAux
#endif
    constexpr int kNt = FP8MI_EPI_AUX;             // aux: streaming (nt) store - C is written once and not re-read here (0 = default policy: A/B builds only)
    const uint32_t ldc_b = (uint32_t)(ldc * kEsz);
    const bool has_sr = (flags & kFSr) != 0;        // uniform: the multiplication is skipped as a block when there is none
    const float sa_u = tab[0], sb_u = tab[128];   // per-tensor: every entry of the table is the scale
    f32x4 nan_sum = {0.0f, 0.0f, 0.0f, 0.0f};
    // One wave per SIMD: nothing but this wave's own instruction order hides the LDS latency.  Round 3: the batches are software-pipelined by
    // hand - the LDS reads of batch t + 1 are issued (and fenced) ahead of the arithmetic and the stores of batch t (as a plain loop hipcc put
    // every trip's reads behind the previous trip's stores: ~200 cycles of exposed latency per trip; with all of an epilogue's stores dropped
    // FLUX only gets 2 % faster, so the epilogue's 11.4 k cycles per tile are its own instruction stream, not the store traffic).
    struct Batch {
        f32x4 q0[kBatch], q1[kBatch], sb0[kBatch], sb1[kBatch], bn0[kBatch], bn1[kBatch];
        float sa[kBatch], bm[kBatch], sa1[kBatch], bm1[kBatch];   // (sa1 / bm1: the second row of the fp32 form)
    };
    auto load = [&](int it0, Batch &B) {
#pragma unroll
        for (int b = 0; b < kBatch; ++b) {
            if constexpr (kF32) {
                const int rrA = (it0 + b) * kRpi + rsubF, rrB = rrA + kRpiF, rA = rrA & 15, rB = rrB & 15;
                lds_cu8 *base = dump + (rrA >> 4) * kDumpRow + P * 16;
                B.q0[b] = *(lds_cf32x4 *)(base + rA * (COLS * 4));
                B.q1[b] = *(lds_cf32x4 *)(base + rB * (COLS * 4));
                const int colA = chunk_col(rA), colB = chunk_col(rB), rowA = half * 64 + rrA, rowB = half * 64 + rrB;
                if (TABLES) {
                    B.sa[b] = tab[rowA];
                    B.sa1[b] = tab[rowB];
                    B.sb0[b] = *(lds_cf32x4 *)(tab + 128 + colA);
                    B.sb1[b] = *(lds_cf32x4 *)(tab + 128 + colB);
                }
                if (BIAS && !TRANSPOSED) {
                    B.bn0[b] = *(lds_cf32x4 *)(tab + 256 + colA);
                    B.bn1[b] = *(lds_cf32x4 *)(tab + 256 + colB);
                }
                if (BIAS && TRANSPOSED) {
                    B.bm[b] = tab[384 + rowA];
                    B.bm1[b] = tab[384 + rowB];
                }
                continue;
            }
            const int rr = (it0 + b) * kRpi + rsub, r = rr & 15;   // row inside the half: fragment row rr >> 4, row r
            lds_cu8 *src = dump + (rr >> 4) * kDumpRow + r * (COLS * 4) + pp * 32;
            const int swap = (r & 1) * 16;                      // odd rows hold the pair's chunks exchanged: undo it in the address
            B.q0[b] = *(lds_cf32x4 *)(src + swap);
            B.q1[b] = *(lds_cf32x4 *)(src + (16 - swap));
            const int col = ((pp ^ (r >> 1)) & (kLpr - 1)) * 8, row = half * 64 + rr;
            if (TABLES) {
                B.sa[b] = tab[row];
                B.sb0[b] = *(lds_cf32x4 *)(tab + 128 + col);
                B.sb1[b] = *(lds_cf32x4 *)(tab + 128 + col + 4);
            }
            if (BIAS && !TRANSPOSED) {
                B.bn0[b] = *(lds_cf32x4 *)(tab + 256 + col);
                B.bn1[b] = *(lds_cf32x4 *)(tab + 256 + col + 4);
            }
            if (BIAS && TRANSPOSED) B.bm[b] = tab[384 + row];
        }
    };
    auto process = [&](int it0, const Batch &B) {
        float v[kBatch][8];
#pragma unroll
        for (int b = 0; b < kBatch; ++b) {
            nan_sum += B.q0[b] + B.q1[b];
            const f32x4 lo = B.q0[b], hi = B.q1[b];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = j < 4 ? lo[j & 3] : hi[j & 3];
                const float s_a = TABLES ? ((kF32 && j >= 4) ? B.sa1[b] : B.sa[b]) : sa_u;
                const float s_b = TABLES ? (j < 4 ? B.sb0[b][j & 3] : B.sb1[b][j & 3]) : sb_u;
                float x = TRANSPOSED ? (a * s_b) * s_a : (a * s_a) * s_b;
                if (BIAS) x = x + (TRANSPOSED ? ((kF32 && j >= 4) ? B.bm1[b] : B.bm[b]) : (j < 4 ? B.bn0[b][j & 3] : B.bn1[b][j & 3]));
                v[b][j] = x;
            }
        }
        if (has_sr) {
#pragma unroll
            for (int b = 0; b < kBatch; ++b)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[b][j] = v[b][j] * sr;
        }
#pragma unroll
        for (int b = 0; b < kBatch; ++b) {
            if constexpr (kF32) {
                const int rrA = (it0 + b) * kRpi + rsubF, rrB = rrA + kRpiF;
                const int colA = chunk_col(rrA & 15), colB = chunk_col(rrB & 15);
                const int offA = colA < cols_ok ? (int)((uint32_t)(half * 64 + rrA) * ldc_b + (uint32_t)(colA * 4)) : 0x7FFFFFF0;
                const int offB = colB < cols_ok ? (int)((uint32_t)(half * 64 + rrB) * ldc_b + (uint32_t)(colB * 4)) : 0x7FFFFFF0;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[b][0], v[b][1], v[b][2], v[b][3]}), rc, offA, 0, kNt);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v[b][4], v[b][5], v[b][6], v[b][7]}), rc, offB, 0, kNt);
                continue;
            }
            const int rr = (it0 + b) * kRpi + rsub, r = rr & 15;
            const int col = ((pp ^ (r >> 1)) & (kLpr - 1)) * 8, row = half * 64 + rr;
            // columns of a ragged last n-tile beyond N: the store's offset is pushed out of the descriptor's range (dropped)
            const int off = col < cols_ok ? (int)((uint32_t)row * ldc_b + (uint32_t)(col * kEsz)) : 0x7FFFFFF0;
            {
                uint32_t w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {   // one packed convert (round to nearest even) per output pair
                    typedef float f32x2_t __attribute__((ext_vector_type(2)));
                    const f32x2_t pr = {v[b][2 * j], v[b][2 * j + 1]};
                    if (OUT == FP8MI_BF16) {
                        typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                        w[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, bf16x2_t));
                    } else {
                        typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
                        w[j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(pr, f16x2_t));
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{w[0], w[1], w[2], w[3]}, rc, off, 0, kNt);
            }
        }
    };
    constexpr int kTrips = kIters / kBatch;
    static_assert(kIters % kBatch == 0 && kTrips >= 1, "whole batches");
    constexpr bool kPipe = !TABLES && kTrips >= 2;   // (with per-row scale tables two batches in flight do not fit the ~118 registers the asm leaves)
    if constexpr (kPipe) {
        Batch A, Bq;
        load(0, A);
#pragma unroll
        for (int t = 0; t < kTrips; t += 2) {
            if (t + 1 < kTrips) load((t + 1) * kBatch, Bq);
            __builtin_amdgcn_sched_barrier(0);   // the next batch's reads stay ABOVE this batch's arithmetic and stores
            process(t * kBatch, A);
            if (t + 1 < kTrips) {
                if (t + 2 < kTrips) load((t + 2) * kBatch, A);
                __builtin_amdgcn_sched_barrier(0);
                process((t + 1) * kBatch, Bq);
            }
        }
    } else {
#pragma unroll 1
        for (int t = 0; t < kTrips; ++t) {
            Batch A;
            load(t * kBatch, A);
            process(t * kBatch, A);
        }
    }
    return nan_sum;
}

template <int COLS, int OUT, bool TABLES>
FP8MI_DEVICE f32x4 epilogue_half_flags(const MMParams &p, int flags, float sr, lds_cu8 *dump, lds_cf32 *tab, int half, int64_t m_wave,
                                       int64_t n_wave, int64_t ldc, int rows_ok, int cols_ok, int lane)
{
    if (!(flags & kFBias)) {
        if (flags & kFTransposed) return epilogue_half<COLS, OUT, false, true, TABLES>(p, flags, sr, dump, tab, half, m_wave, n_wave, ldc, rows_ok, cols_ok, lane);
        return epilogue_half<COLS, OUT, false, false, TABLES>(p, flags, sr, dump, tab, half, m_wave, n_wave, ldc, rows_ok, cols_ok, lane);
    }
    if (flags & kFTransposed) return epilogue_half<COLS, OUT, true, true, TABLES>(p, flags, sr, dump, tab, half, m_wave, n_wave, ldc, rows_ok, cols_ok, lane);
    return epilogue_half<COLS, OUT, true, false, TABLES>(p, flags, sr, dump, tab, half, m_wave, n_wave, ldc, rows_ok, cols_ok, lane);
}

template <int COLS>
FP8MI_DEVICE f32x4 epilogue_half_any(const MMParams &p, int flags, float sr, lds_cu8 *dump, lds_cf32 *tab, int half, int64_t m_wave,
                                     int64_t n_wave, int64_t ldc, int rows_ok, int cols_ok, int lane)
{
    const bool tables = (flags & (kFSaRow | kFSbRow)) != 0;
    const int od = (flags >> kFOutShift) & 3;
    if (od == FP8MI_F32)
        return tables ? epilogue_half_flags<COLS, FP8MI_F32, true>(p, flags, sr, dump, tab, half, m_wave, n_wave, ldc, rows_ok, cols_ok, lane)
                      : epilogue_half_flags<COLS, FP8MI_F32, false>(p, flags, sr, dump, tab, half, m_wave, n_wave, ldc, rows_ok, cols_ok, lane);
    if (od == FP8MI_BF16)
        return tables ? epilogue_half_flags<COLS, FP8MI_BF16, true>(p, flags, sr, dump, tab, half, m_wave, n_wave, ldc, rows_ok, cols_ok, lane)
                      : epilogue_half_flags<COLS, FP8MI_BF16, false>(p, flags, sr, dump, tab, half, m_wave, n_wave, ldc, rows_ok, cols_ok, lane);
    return tables ? epilogue_half_flags<COLS, FP8MI_F16, true>(p, flags, sr, dump, tab, half, m_wave, n_wave, ldc, rows_ok, cols_ok, lane)
                  : epilogue_half_flags<COLS, FP8MI_F16, false>(p, flags, sr, dump, tab, half, m_wave, n_wave, ldc, rows_ok, cols_ok, lane);
}

#ifdef FP8MI_STAMP  // diagnostic build only: phase stamps of wave 0 of every workgroup (tools/stamp_gemm256.py)
__device__ unsigned long long g_stamp256[1024 * 8];
#define STAMP256(i)                                                                 \
    do {                                                                            \
        unsigned long long t_;                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");  \
        if (threadIdx.x == 0 && blockIdx.x < 1024) g_stamp256[blockIdx.x * 8 + (i)] = t_; \
    } while (0)
#else
#define STAMP256(i) do { } while (0)
#endif

template <int V, int BN>   // V = 0: the product schedule; others only in the diagnostic build (gen_gemm256_loop.py VARIANTS, BN = 256)
__global__ __launch_bounds__(kThreads256) void gemm256_kernel(MMParams p_in, int tiles_m, int tiles_n, int nwg)
{
#ifdef FP8MI_STAMP
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_stamp256[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memrealtime();
#endif
    STAMP256(0);
    const MMParams p = pin_params(p_in);
    FP8MI_PIN_S(tiles_m); FP8MI_PIN_S(tiles_n); FP8MI_PIN_S(nwg);
    int flags = (p.bias ? kFBias : 0) | (p.transposed ? kFTransposed : 0) | (p.sa_row ? kFSaRow : 0) | (p.sb_row ? kFSbRow : 0) |
                (p.nan_zero ? kFNanZero : 0) | (p.scale_result ? kFSr : 0) | (p.out_dtype << kFOutShift) | (p.bias_dtype << kFBiasTypeShift);
    FP8MI_PIN_S(flags);
    using G = Geo<BN>;
    constexpr int kBN = BN, kSlotBytes = G::kSlotBytes, kRing256 = G::kRing, kDumpWave = G::kDumpWave, kTabBase = G::kTabBase;
    __shared__ __attribute__((aligned(16))) uint8_t smem[kRing256 + kFlagBytes + 4 * kTabBytes];
    if (threadIdx.x == 0) *(__attribute__((address_space(3))) volatile int *)(lds_void *)(smem + kRing256) = 0;  // NaN verdict word (ordered by the K loop's barriers)

    int tile_m, tile_n, kslice, wg;
    tile_of_block(blockIdx.x, nwg, tiles_m, tiles_n, tile_m, tile_n, kslice, wg);
    const int64_t m0 = (int64_t)tile_m * kBM, n0 = (int64_t)tile_n * kBN;
    // Which quadrant of the tile a wave owns and which row groups it stages only has to be a bijection onto 0..3: the id of the
    // SIMD the wave runs on serves - a wave of this kernel owns its SIMD's whole register file (512 entries: the asm's clobber
    // list sees to that), so the four waves of a workgroup sit on four different SIMDs.  Read from HW_ID (bits 5:4) where it is
    // needed instead of being derived from threadIdx.x once: held in an SGPR across the tile loop it was spilled - through a
    // VGPR to scratch - and reloaded with a full vmcnt wait in front of the K loop.
#define FP8MI_SIMD_ID() ((int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4))

    // raw buffer descriptors (stride 0, range-checked) rebased to this tile's first row
    const uint64_t pa = (uint64_t)(p.A + m0 * p.lda), pb = (uint64_t)(p.B + n0 * p.ldb);
    const int64_t rows_a = min(kBM, (int)p.M - (int)m0);   // the last m-tile may be ragged: rows beyond M read as zeros (range check) and are not stored
    const int64_t rows_b = min(kBN, (int)p.N - (int)n0);   // ... and so may the last n-tile
    const int64_t bytes_a = (rows_a - 1) * p.lda + p.K, bytes_b = (rows_b - 1) * p.ldb + p.K;
    u32x4 ra = {(uint32_t)pa, (uint32_t)(pa >> 32) & 0xFFFFu, (uint32_t)min(bytes_a, (int64_t)0x7FFFFFFF), 0x00020000u};
    u32x4 rb = {(uint32_t)pb, (uint32_t)(pb >> 32) & 0xFFFFu, (uint32_t)min(bytes_b, (int64_t)0x7FFFFFFF), 0x00020000u};
    ra[0] = __builtin_amdgcn_readfirstlane(ra[0]); ra[1] = __builtin_amdgcn_readfirstlane(ra[1]);
    ra[2] = __builtin_amdgcn_readfirstlane(ra[2]);
    rb[0] = __builtin_amdgcn_readfirstlane(rb[0]); rb[1] = __builtin_amdgcn_readfirstlane(rb[1]);
    rb[2] = __builtin_amdgcn_readfirstlane(rb[2]);

    const uint32_t sa = (uint32_t)(32 * p.lda), sb = (uint32_t)(32 * p.ldb);   // byte stride between a wave's consecutive row groups
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void *)smem;
    const uint32_t vscale = (uint32_t)kScaleOne;
    const int nk = (int)((p.K + BK - 1) / BK);  // >= 2 (host); >= 3 when the last K-step is partial
    const int ktail = (int)(p.K % BK);          // bytes of a partial last K-step (a multiple of 16), 0 = none: its stage is staged with per-lane K masks

    STAMP256(1);
    // NaN bytes: the epilogue sums the accumulators it reads anyway; a NaN proves a NaN byte took part in this tile (finite
    // e4m3 products cannot overflow fp32).  Only then - reference semantics, fp8_matmul.metal:21 - the workgroup redoes
    // the tile with every fragment scrubbed and stores it again; clean inputs pay one barrier.
    typedef __attribute__((address_space(3))) volatile int lds_vint;
    lds_vint *flag = (lds_vint *)(lds_void *)(smem + kRing256);
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
        // Operands of the generated K loops, derived from a lane id that an opaque statement produces RIGHT HERE: visible as values
        // shared between the branches below (or as loop invariants), hipcc hoists the ~20 registers of asm operands above the branch,
        // keeps a private copy per statement of every read-write one and spills them around the K loop (7-12 VGPRs with three
        // statements sharing one set; tools/check_spills.py).  Every branch that runs a generated loop expands its own copy.
        //   staging plan (fp8mi_gemm.hip): wave w stages the 1-KiB groups w, w + 4, ... of both operands; lane -> (row, swizzled chunk)
        //   fragment reads: row r = lane & 15, lane group g reads chunks g and 4 + g of its row, swizzled by (r >> 1)
        //   L2 prefetch (gen_gemm256_loop.py pf_group): the 32 tiles an XCD runs at one time are 4 m-tiles x 8 n-tiles of one group
        //   (tile_of_block), so an A panel has 8 readers and a B panel 4; each warms its share of the lines of a later stage: wave 0
        //   rows 64 (tile_m & 3) .. + 63 of its B panel, wave 1 rows 32 (tile_n & 7) .. + 31 of its A panel (other lanes point outside
        //   the buffer: no access; diagnostic variants 19-21 split both shares over the four waves)
        //   accumulator dump: row fr of each fragment row, 16-byte chunk (4 tn + fg) ^ fr
#define FP8MI_G256_SETUP() \
        int lane_l; \
        int fl_l = flags; \
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_l), "+s"(fl_l)); \
        const int wave = FP8MI_SIMD_ID(); \
        const int wm0 = (wave & 1) * 128, wn0 = (wave >> 1) * G::kCols; \
        const int row0 = wave * 8 + (lane_l >> 3); \
        const int chunk = (lane_l & 7) ^ (((wave & 1) * 4 + (lane_l >> 4)) & 7); \
        const uint32_t va0 = (uint32_t)(row0 * p.lda + chunk * 16), vb0 = (uint32_t)(row0 * p.ldb + chunk * 16); \
        const uint32_t va0t = (ktail == 0 || chunk * 16 < ktail) ? va0 : kOOB, vb0t = (ktail == 0 || chunk * 16 < ktail) ? vb0 : kOOB; \
        const int fr = lane_l & 15, fg = lane_l >> 4; \
        const uint32_t off1 = (uint32_t)(fr * BK + ((fg ^ (fr >> 1)) << 4)), off2 = (uint32_t)(fr * BK + (((4 + fg) ^ (fr >> 1)) << 4)); \
        uint32_t alo_c = lds0 + wm0 * BK + off1, ahi_c = lds0 + wm0 * BK + off2; \
        uint32_t blo_c = lds0 + kBM * BK + wn0 * BK + off1, bhi_c = lds0 + kBM * BK + wn0 * BK + off2; \
        uint32_t alo_n = alo_c + kSlotBytes, ahi_n = ahi_c + kSlotBytes, blo_n = blo_c + kSlotBytes, bhi_n = bhi_c + kSlotBytes; \
        uint32_t m0_c = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + wave * 1024)), m0_n = m0_c + kSlotBytes; \
        uint32_t k2 = 0, nloop = (uint32_t)(nk - 2), t0, t1; \
        constexpr bool kSplitPf = V >= 19 && V <= 21; \
        uint32_t pfoff; \
        if (!kSplitPf) { \
            pfoff = (wave == 0 && lane_l < BN / 4) ? (uint32_t)((BN / 4 * (tile_m & 3) + lane_l) * p.ldb) \
                              : (wave == 1 && lane_l < 32) ? (uint32_t)((32 * (tile_n & 7) + lane_l) * p.lda) : kOOB; \
        } else { \
            pfoff = lane_l < 16 ? (uint32_t)((64 * (tile_m & 3) + 16 * wave + lane_l) * p.ldb) : kOOB; \
        } \
        const uint32_t wave_s = (uint32_t)wave; \
        const u32x4 rpf = (kSplitPf || wave == 0) ? rb : ra; \
        const uint32_t klast = (uint32_t)((nk - 1) * BK); \
        const uint32_t drow = lds0 + wave * kDumpWave + fr * (G::kCols * 4), dkey = (uint32_t)((fg ^ fr) << 4);
        f32x4 t;
        int fl_o = flags;
        asm volatile("" : "+s"(fl_o));
        (void)fl_o;
        // Fused tail (diagnostic library only, FP8MI_DEBUG bit 2; gen_gemm256_loop.py fused_last_step): the epilogue of the common case -
        // per-tensor scales, no bias / scale_result, tile whole in N - runs under the last K-step's MFMAs straight from the AGPRs.  Built,
        // bit-identical, and SLOWER than the LDS dump (FLUX 120.7 vs 118.5 us): see the generator's note and profiles/r03_tail_probe.txt.
#ifdef FP8MI_DIAG
        const bool fused = (p.debug & 4) && V == 0 && pass == 0 && !(fl_o & (kFBias | kFSaRow | kFSbRow | kFSr)) && (n0 + kBN <= p.N);
        if (fused) {
            FP8MI_G256_SETUP();
            const int fl_u = __builtin_amdgcn_readfirstlane(fl_l);   // (hipcc treats the results of a two-output asm statement as divergent: the scalar operands below need SGPRs)
            const int od = (fl_u >> kFOutShift) & 3, esz = od == FP8MI_F32 ? 4 : 2;
            const int64_t m_wave = m0 + wm0, n_wave = n0 + wn0;
            const int rows_ok = min(max((int)p.M - (int)m_wave, 0), 128);   // rows of a ragged last m-tile beyond M: out of the descriptor's range, dropped
            const uint64_t pc = (uint64_t)((uint8_t *)p.C + (m_wave * p.ldc + n_wave) * esz);
            const uint32_t ldc_b = (uint32_t)(p.ldc * esz);
            u32x4 frc = {(uint32_t)pc, (uint32_t)(pc >> 32) & 0xFFFFu, (uint32_t)rows_ok * ldc_b, 0x00020000u};
            frc[0] = __builtin_amdgcn_readfirstlane(frc[0]); frc[1] = __builtin_amdgcn_readfirstlane(frc[1]);
            frc[2] = __builtin_amdgcn_readfirstlane(frc[2]);
            // lane (fr, fg) ends up with the 8 consecutive columns (fg & 1) * 16 + (fg >> 1) * 8 ..+7 of a fragment pair, row fr of row block tm
            const uint32_t fvoff = (uint32_t)fr * ldc_b + (uint32_t)(((fg & 1) * 16 + (fg >> 1) * 8) * esz);
            const uint32_t fsrow = 16u * ldc_b;
            // (acc * s1) * s2: scale_a first, scale_b first in the transposed epilogue - the ring kernels' order (fp8mi_gemm_epi.h)
            const float *fpm1 = (fl_u & kFTransposed) ? p.scale_b : p.scale_a, *fpm2 = (fl_u & kFTransposed) ? p.scale_a : p.scale_b;
            float nanv;
            uint32_t fs1, fs2;
            const uint32_t fod = (uint32_t)od;
            if constexpr (BN == 128) {
                FP8MI_GEMM256_LOOP_FUSED_N128();
            } else {
                FP8MI_GEMM256_LOOP_FUSED();
            }
            (void)t0; (void)t1; (void)fs1; (void)fs2;
            t = f32x4{nanv, 0.0f, 0.0f, 0.0f};
            STAMP256(2); STAMP256(3);
        } else
#endif
        {
        FP8MI_G256_SETUP();
        const TabRegs tabs = load_tables(p, fl_l, m0 + wm0, n0 + wn0, G::kCols, lane_l);   // in flight under the K loop, stored to the LDS behind it
        f32x32 acc4, acc5, acc6, acc7;   // fragment rows 4..7, pinned by the asm to a[128:255] (BN = 256) / a[64:127] (BN = 128: acc4, acc5)
        if (pass == 0) {
            if constexpr (BN == 128 && V == 0) {
                FP8MI_GEMM256_LOOP_N128();
            } else if constexpr (V == 0) {
                FP8MI_GEMM256_LOOP();
            }
#ifdef FP8MI_DIAG
#define X(v) else if constexpr (BN == 256 && V == v) { FP8MI_GEMM256_LOOP_V##v(); }
            FP8MI_GEMM256_VARIANTS
#undef X
#define X(v) else if constexpr (BN == 128 && V == v) { FP8MI_GEMM256_LOOP_N128_V##v(); }
            FP8MI_GEMM256_NVARIANTS
#undef X
#endif
        } else {
            uint32_t vt0, vt1;
            if constexpr (BN == 128) {
                FP8MI_GEMM256_LOOP_SCRUB_N128();
            } else {
                FP8MI_GEMM256_LOOP_SCRUB();
            }
            (void)vt0; (void)vt1;
        }
        (void)t0; (void)t1;
        if (pass == 0) STAMP256(2);
        int lane_e;
        const int wave_e = FP8MI_SIMD_ID();   // (read again: nothing of the wave's identity is kept across the K loop)
        int64_t m_wave = m0 + (wave_e & 1) * 128, n_wave = n0 + (wave_e >> 1) * G::kCols;
        int fl = flags;        // (the switches and ldc too: their tests and multiples are then formed here, not in SGPRs held across the loop)
        int64_t ldc_e = p.ldc;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_e), "+s"(m_wave), "+s"(n_wave), "+s"(fl), "+s"(ldc_e));
        const int cols_ok = min(max((int)p.N - (int)n_wave, 0), G::kCols);   // ... and columns (a multiple of the store width: fp8mi_gemm256_supported)
        int rows_ok = min(max((int)p.M - (int)m_wave, 0), 128);   // valid rows of this wave's tile (ragged last m-tile)
#ifdef FP8MI_DIAG
        if (p.debug & 8) rows_ok = 0;   // timing-only: every store of the epilogue falls outside its descriptor and is dropped (what the epilogue costs without its store traffic)
#endif
        lds_cu8 *dump = (lds_cu8 *)(lds_void *)(smem + wave_e * kDumpWave);
        float *tabw = (float *)(smem + kTabBase + wave_e * kTabBytes);
        store_tables(fl, tabs, tabw, lane_e);
        lds_cf32 *tab = (lds_cf32 *)(lds_void *)tabw;
        const float sr = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tabw[512])));
        t = epilogue_half_any<G::kCols>(p, fl, sr, dump, tab, 0, m_wave, n_wave, ldc_e, rows_ok, cols_ok, lane_e);
        if (pass == 0) STAMP256(3);
        if constexpr (BN == 128) {
            FP8MI_GEMM256_DUMP_HI_N128();
        } else {
            FP8MI_GEMM256_DUMP_HI();
        }
        (void)acc6; (void)acc7;
        t += epilogue_half_any<G::kCols>(p, fl, sr, dump, tab, 1, m_wave, n_wave, ldc_e, rows_ok, cols_ok, lane_e);
        }
        if (!(flags & kFNanZero) || pass == 1) break;
        const float sum = (t[0] + t[1]) + (t[2] + t[3]);
        if (sum != sum) *flag = 1;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // every wave has voted (and is done with its dump)
        if (!__builtin_amdgcn_readfirstlane(*flag)) break;  // workgroup-uniform
    }
#ifdef FP8MI_STAMP
    STAMP256(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP256(5);
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_stamp256[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memrealtime();
#endif
}

}  // namespace

// any M and N (ragged last tiles; N a multiple of the 16-byte store: 4 fp32 / 8 half columns), K >= 256 (a multiple of 16; round 3: a partial last
// K-step is staged with per-lane masks by the loop's peeled step), 16-byte aligned output rows, no split-K
bool fp8mi_gemm256_supported(const MMParams &p)
{
    const int esz = p.out_dtype == FP8MI_F32 ? 4 : 2;
    // (K: a multiple of 16 as for every tile kernel; whole K-steps need two of them, a partial last one three - the tail stage is staged by the peeled step)
    return fp8mi_gemm_supported(p) && p.M < 0x7FFFFF00 && p.N < 0x7FFFFF00 && (p.N % (16 / esz)) == 0 && ((p.K % BK) == 0 ? p.K >= 2 * BK : p.K > 2 * BK) && p.split <= 1 &&
           ((p.ldc * esz) % 16) == 0 && (((uintptr_t)p.C) % 16) == 0 && ((p.M + kBM - 1) / kBM) * ((p.N + 127) / 128) <= 0x7FFFFFFF &&
           p.ldc * esz * 128 < 0x7FFF0000;   // the epilogue addresses a wave tile (128 rows) with 32-bit offsets
}

#ifdef FP8MI_STAMP
extern "C" int fp8mi_debug_read_stamps256(unsigned long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp256), sizeof(unsigned long long) * n);
}
#endif

// variant: 0 = the product loop on 256x256 tiles, 1000 = the same schedule on 256x128 tiles; others (diagnostic build) = schedule variants
int fp8mi_launch_gemm256(const MMParams &p, int variant, hipStream_t s)
{
    const int bn = variant >= 1000 ? 128 : 256;
    const int64_t tm = (p.M + kBM - 1) / kBM, tn = (p.N + bn - 1) / bn;
    if (tm * tn > 0x7FFFFFFF) return FP8MI_E_UNSUPPORTED;
    const unsigned grid = (unsigned)(tm * tn);
    switch (variant) {
    case 0: return fp8mi_launch(gemm256_kernel<0, 256>, dim3(grid), dim3(kThreads256), s, p, (int)tm, (int)tn, (int)grid);
    case 1000: return fp8mi_launch(gemm256_kernel<0, 128>, dim3(grid), dim3(kThreads256), s, p, (int)tm, (int)tn, (int)grid);
#ifdef FP8MI_DIAG
#define X(v) case v: return fp8mi_launch(gemm256_kernel<v, 256>, dim3(grid), dim3(kThreads256), s, p, (int)tm, (int)tn, (int)grid);
    FP8MI_GEMM256_VARIANTS
#undef X
#define X(v) case 1000 + v: return fp8mi_launch(gemm256_kernel<v, 128>, dim3(grid), dim3(kThreads256), s, p, (int)tm, (int)tn, (int)grid);
    FP8MI_GEMM256_NVARIANTS
#undef X
#endif
    default: return FP8MI_E_ENUM;
    }
}
