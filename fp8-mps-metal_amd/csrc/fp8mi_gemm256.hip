// 256x256-tile FP8 e4m3fn GEMM with ONE wave per SIMD and a hand-scheduled K loop (gfx950).
//
// Same contract and the same bits as the ring kernels of fp8mi_gemm.hip (same LDS image, same fragment -> MFMA operand
// map, K-steps added in order), for the shapes that are made of whole 256x256 tiles and whole 128-byte K-steps:
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

namespace {

constexpr int kBM = 256, kBN = 256, kThreads256 = 256;   // 4 waves, one per SIMD
constexpr int kSlotBytes = (kBM + kBN) * BK;   // 64 KiB: A's 256 rows, then B's
constexpr int kRing256 = 2 * kSlotBytes;

// one fragment row (16 rows x 128 columns of the wave tile) through the fused epilogue, staged through the wave's corner
// of the idle ring so that every global store writes whole lines (cf. epilogue_staged, fp8mi_gemm_epi.h)
template <int OUT>
FP8MI_DEVICE void epilogue_row(const MMParams &p, const EpiScalars &es, const f32x4 (&acc)[8], const float (&sbv)[8][4],
                               const float (&bv)[8][4], uint8_t *buf, int64_t row0 /* global m of the fragment row */,
                               int64_t col0 /* global n of the wave tile */, int lane)
{
    constexpr int kEsz = OUT == FP8MI_F32 ? 4 : 2;
    constexpr int kRowBytes = 128 * kEsz, kStride = kRowBytes + 16, kCPR = kRowBytes / 16, kRPI = 64 / kCPR, kNI = 16 / kRPI;
    const int fr = lane & 15, fg = lane >> 4;
    const bool has_bias = p.bias != nullptr, has_sr = p.scale_result != nullptr;
    const float sa = p.sa_row ? p.scale_a[row0 + fr] : es.sa0;
    const float brow = (has_bias && p.transposed) ? load_as_float(p.bias, row0 + fr, p.bias_dtype) : 0.0f;
#pragma unroll
    for (int tn = 0; tn < 8; ++tn) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float r = p.transposed ? (acc[tn][j] * sbv[tn][j]) * sa : (acc[tn][j] * sa) * sbv[tn][j];
            if (has_bias) r = r + (p.transposed ? brow : bv[tn][j]);
            if (has_sr) r = r * es.sr;
            v[j] = r;
        }
        uint8_t *d = buf + fr * kStride + (tn * 16 + fg * 4) * kEsz;
        if (OUT == FP8MI_F32) {
            *(f32x4 *)d = f32x4{v[0], v[1], v[2], v[3]};
        } else if (OUT == FP8MI_BF16) {
            __bf16 h0 = (__bf16)v[0], h1 = (__bf16)v[1], h2 = (__bf16)v[2], h3 = (__bf16)v[3];
            *(u32x2 *)d = u32x2{(uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16),
                                (uint32_t)__builtin_bit_cast(uint16_t, h2) | ((uint32_t)__builtin_bit_cast(uint16_t, h3) << 16)};
        } else {
            _Float16 h0 = (_Float16)v[0], h1 = (_Float16)v[1], h2 = (_Float16)v[2], h3 = (_Float16)v[3];
            *(u32x2 *)d = u32x2{(uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16),
                                (uint32_t)__builtin_bit_cast(uint16_t, h2) | ((uint32_t)__builtin_bit_cast(uint16_t, h3) << 16)};
        }
    }
    const int rrow = lane / kCPR, rchunk = lane % kCPR;
    uint8_t *grow = (uint8_t *)p.C + (row0 * p.ldc + col0) * kEsz;
#pragma unroll
    for (int i = 0; i < kNI; ++i) {
        const int r = i * kRPI + rrow;
        u32x4 q = *(const u32x4 *)(buf + r * kStride + rchunk * 16);  // same wave wrote and reads: DS operations of one wave execute in order
        __builtin_nontemporal_store(q, (u32x4 *)(grow + (int64_t)r * p.ldc * kEsz + rchunk * 16));
    }
}

template <int OUT, int TM>
FP8MI_DEVICE void epilogue_rows(const MMParams &p, const EpiScalars &es, const float (&sbv)[8][4], const float (&bv)[8][4],
                                uint8_t *buf, int64_t m_wave, int64_t n_wave, int lane)
{
    if constexpr (TM < 8) {
        f32x4 r[8];
        read_acc_row<TM>(r);
        epilogue_row<OUT>(p, es, r, sbv, bv, buf, m_wave + TM * 16, n_wave, lane);
        epilogue_rows<OUT, TM + 1>(p, es, sbv, bv, buf, m_wave, n_wave, lane);
    }
}

template <int TM>
FP8MI_DEVICE void sum_acc_rows(f32x4 &t)
{
    if constexpr (TM < 8) {
        f32x4 r[8];
        read_acc_row<TM>(r);
#pragma unroll
        for (int tn = 0; tn < 8; ++tn) t += r[tn];
        sum_acc_rows<TM + 1>(t);
    }
}

template <int OUT>
FP8MI_DEVICE void epilogue256(const MMParams &p, const EpiScalars &es, uint8_t *smem, int64_t m0, int64_t n0, int wave, int wm0,
                              int wn0, int lane)
{
    constexpr int kEsz = OUT == FP8MI_F32 ? 4 : 2;
    constexpr int kStride = 128 * kEsz + 16;
    uint8_t *buf = smem + wave * (16 * kStride);
    const int fg = lane >> 4;
    const bool has_bias = p.bias != nullptr;
    float sbv[8][4], bv[8][4];
#pragma unroll
    for (int tn = 0; tn < 8; ++tn)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = n0 + wn0 + tn * 16 + fg * 4 + j;
            sbv[tn][j] = p.sb_row ? p.scale_b[n] : es.sb0;
            bv[tn][j] = (has_bias && !p.transposed) ? load_as_float(p.bias, n, p.bias_dtype) : 0.0f;
        }
    epilogue_rows<OUT, 0>(p, es, sbv, bv, buf, m0 + wm0, n0 + wn0, lane);
}

__global__ __launch_bounds__(kThreads256) void gemm256_kernel(MMParams p_in, int tiles_m, int tiles_n, int nwg)
{
    const MMParams p = pin_params(p_in);
    FP8MI_PIN_S(tiles_m); FP8MI_PIN_S(tiles_n); FP8MI_PIN_S(nwg);
    const EpiScalars es = load_epi_scalars(p);
    __shared__ __attribute__((aligned(16))) uint8_t smem[kRing256 + kFlagBytes];
    if (threadIdx.x == 0) *(volatile int *)(smem + kRing256) = 0;  // NaN verdict word (ordered by the K loop's barriers)

    int tile_m, tile_n, kslice, wg;
    tile_of_block(blockIdx.x, nwg, tiles_m, tiles_n, tile_m, tile_n, kslice, wg);
    const int64_t m0 = (int64_t)tile_m * kBM, n0 = (int64_t)tile_n * kBN;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm0 = (wave & 1) * 128, wn0 = (wave >> 1) * 128;

    // raw buffer descriptors (stride 0, range-checked) rebased to this tile's first row; whole tiles only
    const uint64_t pa = (uint64_t)(p.A + m0 * p.lda), pb = (uint64_t)(p.B + n0 * p.ldb);
    const int64_t bytes_a = (int64_t)(kBM - 1) * p.lda + p.K, bytes_b = (int64_t)(kBN - 1) * p.ldb + p.K;
    u32x4 ra = {(uint32_t)pa, (uint32_t)(pa >> 32) & 0xFFFFu, (uint32_t)min(bytes_a, (int64_t)0x7FFFFFFF), 0x00020000u};
    u32x4 rb = {(uint32_t)pb, (uint32_t)(pb >> 32) & 0xFFFFu, (uint32_t)min(bytes_b, (int64_t)0x7FFFFFFF), 0x00020000u};
    ra[0] = __builtin_amdgcn_readfirstlane(ra[0]); ra[1] = __builtin_amdgcn_readfirstlane(ra[1]);
    ra[2] = __builtin_amdgcn_readfirstlane(ra[2]);
    rb[0] = __builtin_amdgcn_readfirstlane(rb[0]); rb[1] = __builtin_amdgcn_readfirstlane(rb[1]);
    rb[2] = __builtin_amdgcn_readfirstlane(rb[2]);

    // staging plan (fp8mi_gemm.hip): wave w stages the 1-KiB groups w, w + 4, ... of both operands; lane -> (row, swizzled chunk)
    const int row0 = wave * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ (((wave & 1) * 4 + (lane >> 4)) & 7);
    const uint32_t va0 = (uint32_t)(row0 * p.lda + chunk * 16), vb0 = (uint32_t)(row0 * p.ldb + chunk * 16);
    const uint32_t sa = (uint32_t)__builtin_amdgcn_readfirstlane((int)(32 * p.lda)), sb = (uint32_t)__builtin_amdgcn_readfirstlane((int)(32 * p.ldb));
    // fragment read addresses: row r = lane & 15, lane group g reads chunks g and 4 + g of its row, swizzled by (r >> 1)
    const int fr = lane & 15, fg = lane >> 4;
    const uint32_t off1 = (uint32_t)(fr * BK + ((fg ^ (fr >> 1)) << 4)), off2 = (uint32_t)(fr * BK + (((4 + fg) ^ (fr >> 1)) << 4));
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void *)smem;
    const uint32_t vscale = (uint32_t)kScaleOne;
    const int nk = (int)(p.K / BK);  // >= 2 (host)

    auto run = [&](bool scrub) {
        uint32_t alo_c = lds0 + wm0 * BK + off1, ahi_c = lds0 + wm0 * BK + off2;
        uint32_t blo_c = lds0 + kBM * BK + wn0 * BK + off1, bhi_c = lds0 + kBM * BK + wn0 * BK + off2;
        uint32_t alo_n = alo_c + kSlotBytes, ahi_n = ahi_c + kSlotBytes, blo_n = blo_c + kSlotBytes, bhi_n = bhi_c + kSlotBytes;
        uint32_t m0_c = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds0 + wave * 1024)), m0_n = m0_c + kSlotBytes;
        uint32_t k2 = 0, nloop = (uint32_t)(nk - 2), t0;
        if (!scrub) {
            FP8MI_GEMM256_LOOP();
        } else {
            uint32_t vt0, vt1;
            FP8MI_GEMM256_LOOP_SCRUB();
            (void)vt0; (void)vt1;
        }
        (void)t0;
    };
    run(false);

    // ---- end of the K loop: one barrier frees the ring and carries the NaN verdict ----
    volatile int *flag = (volatile int *)(smem + kRing256);
    if (p.nan_zero) {
        f32x4 t = {0.0f, 0.0f, 0.0f, 0.0f};
        sum_acc_rows<0>(t);
        const float s = (t[0] + t[1]) + (t[2] + t[3]);
        if (s != s) *flag = 1;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (p.nan_zero && *flag) {  // workgroup-uniform
        run(true);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    if (p.out_dtype == FP8MI_F32) epilogue256<FP8MI_F32>(p, es, smem, m0, n0, wave, wm0, wn0, lane);
    else if (p.out_dtype == FP8MI_BF16) epilogue256<FP8MI_BF16>(p, es, smem, m0, n0, wave, wm0, wn0, lane);
    else epilogue256<FP8MI_F16>(p, es, smem, m0, n0, wave, wm0, wn0, lane);
}

}  // namespace

// whole 256x256 tiles, whole K-steps (at least two), rows the vector epilogue can store, no split-K
bool fp8mi_gemm256_supported(const MMParams &p)
{
    const int esz = p.out_dtype == FP8MI_F32 ? 4 : 2;
    return fp8mi_gemm_supported(p) && (p.M % kBM) == 0 && (p.N % kBN) == 0 && (p.K % BK) == 0 && p.K >= 2 * BK && p.split <= 1 &&
           ((p.ldc * esz) % 16) == 0 && (((uintptr_t)p.C) % 16) == 0 && (p.M / kBM) * (p.N / kBN) <= 0x7FFFFFFF;
}

int fp8mi_launch_gemm256(const MMParams &p, hipStream_t s)
{
    const int64_t tm = p.M / kBM, tn = p.N / kBN;
    const unsigned grid = (unsigned)(tm * tn);
    return fp8mi_launch(gemm256_kernel, dim3(grid), dim3(kThreads256), s, p, (int)tm, (int)tn, (int)grid);
}
