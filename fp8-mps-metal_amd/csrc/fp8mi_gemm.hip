// LDS-tiled FP8 e4m3fn GEMM for gfx950 on the CDNA4 scaled-MFMA path:
//
//     C[m,n] = cast(((sum_k dec(A[m,k]) dec(B[n,k])) * sa[m] * sb[n] + bias[n]) * sr)
//
// Replaces fp8_scaled_matmul_kernel (fp8_matmul.metal:99-147: one thread per
// output, 2K scalar byte loads and 2K software decodes per output, no reuse)
// and the dequant -> fp16 matmul detour fp8_scaled_mm_fast
// (fp8_mps_native.py:213-267).  Design, MI355X-first:
//
//   * math: v_mfma_scale_f32_16x16x128_f8f6f4 with both block scales = 2^0.
//     It consumes OCP e4m3fn bytes directly (no decode instructions at all) at
//     twice the rate of the non-scaled fp8 MFMA; fp32 accumulators.
//   * both operands are "K-contiguous rows" (A is (M,K), B is (N,K)), so one
//     staging routine and one fragment reader serve both.  The MFMA pairs the
//     j-th byte of lane-group g of its two operands, so any assignment of the
//     128 k-bytes of a step to (g, j) is correct as long as A and B use the
//     same one; we use chunk g and chunk 4+g (16-byte chunks) for lane group
//     g, which with the XOR swizzle below makes every ds_read_b128 of a
//     fragment bank-conflict free.
//   * staging: buffer_load_dwordx4 ... lds (HBM/L2 -> LDS without touching
//     VGPRs).  One wave-instruction moves 8 rows x 128 B = whole cache lines.
//     The LDS image is lane-linear (hardware rule), so the swizzle
//     chunk' = chunk ^ ((row >> 1) & 7) is applied to the per-lane SOURCE
//     address and again on the fragment read.  Rows >= M / N and the K tail
//     are masked by pointing the lane outside the buffer (hardware returns 0,
//     and a zero byte is +0.0 in e4m3).
//   * pipeline: an NSTAGE-deep LDS ring.  NSTAGE-1 K-steps of LDS-DMA stay in
//     flight ACROSS the per-step barrier: the wait is a counted
//     `s_waitcnt vmcnt(n)` (n = loads of the newer stages), the barrier a raw
//     s_barrier - a __syncthreads() would drain the DMA queue (vmcnt(0)) and
//     expose a full memory round trip per K-step (measured on the 2-stage
//     version of this kernel: 49-66 % of wave time parked in waits,
//     profiles/r01_v1_*).  One barrier per K-step.
//   * output orientation: the W fragment is the MFMA "A" operand and the X
//     fragment the "B" operand, so each lane ends up with 4 CONSECUTIVE n of
//     one row m in an accumulator register quad -> one 16-byte store.
//   * the epilogue (scales, bias, result scale, cast) is fused.
//   * block -> tile mapping is XCD-aware: the 8 XCDs get contiguous runs of
//     tiles (m fastest) so that the tiles sharing a B panel hit one L2.
//
// NaN bytes: the hardware treats 0x7F/0xFF as NaN; the reference decodes them
// to 0.0 (fp8_matmul.metal:21).  The K loop runs unscrubbed; because finite
// e4m3 products cannot overflow fp32, a NaN accumulator proves a NaN byte was
// involved, and only then the workgroup redoes its tile with a SWAR scrub of
// every fragment.  Clean inputs (everything the reference's encoder can emit)
// never pay for it.

#include "fp8mi_common.h"

namespace {

constexpr int BK = 128;  // bytes (= k elements) per K-step
constexpr uint32_t kOOB = 0x80000000u;
constexpr int kScaleOne = 0x7F7F7F7F;  // E8M0 127 = 2^0 in every byte

typedef __attribute__((address_space(3))) void lds_void;

template <int BM, int BN, int WM, int WN, int NSTAGE_>
struct Cfg {
    static constexpr int NSTAGE = NSTAGE_;
    static constexpr int PF = NSTAGE_ - 1;  // K-steps of loads in flight
    static constexpr int kWavesM = BM / WM;
    static constexpr int kWavesN = BN / WN;
    static constexpr int kWaves = kWavesM * kWavesN;
    static constexpr int kThreads = kWaves * 64;
    static constexpr int TM = WM / 16;
    static constexpr int TN = WN / 16;
    static constexpr int kGroupsA = BM / 8;  // 8-row staging groups
    static constexpr int kGroupsB = BN / 8;
    static constexpr int kGroups = kGroupsA + kGroupsB;
    static constexpr int kGroupsPerWave = kGroups / kWaves;
    static constexpr int kStageBytes = (BM + BN) * BK;
    static_assert(kGroups % kWaves == 0, "staging groups must divide evenly over the waves");
    static_assert(NSTAGE_ >= 2 && (NSTAGE_ - 2) * kGroupsPerWave <= 63, "vmcnt is a 6-bit counter");
    static_assert(NSTAGE_ * kStageBytes <= 160 * 1024, "LDS is 160 KiB per CU");
};

// one K-step's fragments -> MFMAs
template <typename C, bool SCRUB>
FP8MI_DEVICE void compute_step(const uint8_t *stage, int a_row0 /* m */, int b_row0 /* n */, uint32_t off1,
                               uint32_t off2, f32x4 (&acc)[C::TN][C::TM])
{
    i32x8 xf[C::TM], wf[C::TN];
    const uint8_t *sa = stage + a_row0 * BK;                       // X rows (m) first ...
    const uint8_t *sB = stage + (C::kGroupsA * 8 + b_row0) * BK;   // ... then W rows (n)
#pragma unroll
    for (int t = 0; t < C::TM; ++t) {
        i32x4 lo = *(const i32x4 *)(sa + t * 16 * BK + off1);
        i32x4 hi = *(const i32x4 *)(sa + t * 16 * BK + off2);
        xf[t] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int t = 0; t < C::TN; ++t) {
        i32x4 lo = *(const i32x4 *)(sB + t * 16 * BK + off1);
        i32x4 hi = *(const i32x4 *)(sB + t * 16 * BK + off2);
        wf[t] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
    if (SCRUB) {
#pragma unroll
        for (int t = 0; t < C::TM; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[t][j] = (int)scrub_nan4((uint32_t)xf[t][j]);
#pragma unroll
        for (int t = 0; t < C::TN; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) wf[t][j] = (int)scrub_nan4((uint32_t)wf[t][j]);
    }
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm)
            acc[tn][tm] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[tn], xf[tm], acc[tn][tm], 0, 0, 0,
                                                                            kScaleOne, 0, kScaleOne);
}

template <typename C>
struct StagePlan {
    uint32_t voff[C::kGroupsPerWave];  // per-lane byte offset inside the operand's buffer, or kOOB
    uint32_t kpos[C::kGroupsPerWave];  // chunk*16: position of this lane's 16 bytes inside the K-step
};

template <typename C, bool TAIL>
FP8MI_DEVICE void issue_stage(const StagePlan<C> &pl, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb,
                              uint8_t *stage, int wave, int k0, int64_t K)
{
#pragma unroll
    for (int j = 0; j < C::kGroupsPerWave; ++j) {
        const int gi = wave + j * C::kWaves;  // wave-uniform group index
        uint32_t vo = pl.voff[j];
        if (TAIL && (int64_t)k0 + pl.kpos[j] >= K) vo = kOOB;  // K tail: only in the peeled last step
        // the LDS image is consecutive 1-KiB groups (8 rows x 128 B), A's rows first, then B's
        lds_void *dst = (lds_void *)(stage + gi * 1024);
        if (gi < C::kGroupsA) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, dst, 16, (int)vo, k0, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, dst, 16, (int)vo, k0, 0, 0);
    }
}

template <int N>
FP8MI_DEVICE void wait_loads_and_lds()
{
    // this wave's LDS-DMA except the N youngest have landed; its ds_reads are done
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

template <typename C>
FP8MI_DEVICE void wait_stage(int newer_stages)
{
    constexpr int G = C::kGroupsPerWave;
    if (C::PF >= 2 && newer_stages >= C::PF - 1) wait_loads_and_lds<(C::PF - 1) * G>();
    else if (C::PF >= 5 && newer_stages == 3) wait_loads_and_lds<3 * G>();
    else if (C::PF >= 4 && newer_stages == 2) wait_loads_and_lds<2 * G>();
    else if (C::PF >= 3 && newer_stages == 1) wait_loads_and_lds<1 * G>();
    else wait_loads_and_lds<0>();
}

template <typename C>
FP8MI_DEVICE void issue_any(const StagePlan<C> &pl, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb,
                            uint8_t *stage, int wave, int step, int nk, bool ktail, int64_t K)
{
    if (ktail && step == nk - 1) issue_stage<C, true>(pl, ra, rb, stage, wave, step * BK, K);
    else issue_stage<C, false>(pl, ra, rb, stage, wave, step * BK, K);
}

template <typename C, bool SCRUB>
FP8MI_DEVICE void run_tile(const MMParams &p, uint8_t *smem, const StagePlan<C> &pl, __amdgpu_buffer_rsrc_t ra,
                           __amdgpu_buffer_rsrc_t rb, int wave, int wm0, int wn0, uint32_t off1, uint32_t off2,
                           f32x4 (&acc)[C::TN][C::TM])
{
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm) acc[tn][tm] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const int64_t K = p.K;
    const int nk = (int)((K + BK - 1) / BK);
    const bool ktail = (K % BK) != 0;  // then the last step is staged with per-lane K masking
    if (nk == 0) return;

    // prologue: PF stages in flight (stage s -> ring slot s)
#pragma unroll
    for (int s = 0; s < C::PF; ++s)
        if (s < nk) issue_any<C>(pl, ra, rb, smem + s * C::kStageBytes, wave, s, nk, ktail, K);

    int slot = 0;             // ring slot of step t
    int fill = C::PF % C::NSTAGE;  // ring slot the next issue goes to (= slot of step t-1)
    for (int t = 0; t < nk; ++t) {
        // stage t has landed for this wave once at most the newer stages' loads are outstanding
        wait_stage<C>(min(C::PF - 1, nk - 1 - t));
        __builtin_amdgcn_s_barrier();  // ... for every wave; and every wave is done reading slot of step t-1
        if (t + C::PF < nk) issue_any<C>(pl, ra, rb, smem + fill * C::kStageBytes, wave, t + C::PF, nk, ktail, K);
        compute_step<C, SCRUB>(smem + slot * C::kStageBytes, wm0, wn0, off1, off2, acc);
        slot = (slot + 1 == C::NSTAGE) ? 0 : slot + 1;
        fill = (fill + 1 == C::NSTAGE) ? 0 : fill + 1;
    }
    // all loads were waited for in the last iteration (newer_stages == 0); make the ring reusable
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

template <int BM, int BN, int WM, int WN, int NSTAGE>
__global__ __launch_bounds__((Cfg<BM, BN, WM, WN, NSTAGE>::kThreads)) void gemm_kernel(MMParams p, int tiles_m, int vec_store)
{
    using C = Cfg<BM, BN, WM, WN, NSTAGE>;
    __shared__ __attribute__((aligned(16))) uint8_t smem[NSTAGE * C::kStageBytes];

    // ---- XCD-aware, bijective block -> tile map (m fastest) -------------
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tile_m = wg % tiles_m, tile_n = wg / tiles_m;
    const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm0 = (wave % C::kWavesM) * WM;
    const int wn0 = (wave / C::kWavesM) * WN;

    // ---- buffer descriptors rebased to this tile's first row ------------
    const int64_t rows_a = min((int64_t)BM, p.M - m0), rows_b = min((int64_t)BN, p.N - n0);
    const int64_t bytes_a = (rows_a - 1) * p.lda + p.K, bytes_b = (rows_b - 1) * p.ldb + p.K;
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)(p.A + m0 * p.lda), 0,
                                                                   (int)min(bytes_a, (int64_t)0x7FFFFFFF), 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)(p.B + n0 * p.ldb), 0,
                                                                   (int)min(bytes_b, (int64_t)0x7FFFFFFF), 0x00020000);

    // ---- per-lane staging plan (loop invariant) --------------------------
    StagePlan<C> pl;
#pragma unroll
    for (int j = 0; j < C::kGroupsPerWave; ++j) {
        const int gi = wave + j * C::kWaves;
        const bool is_a = gi < C::kGroupsA;
        const int row = (is_a ? gi : gi - C::kGroupsA) * 8 + (lane >> 3);  // row inside the tile
        const int chunk = (lane & 7) ^ ((row >> 1) & 7);                  // source-side swizzle
        const int64_t rows = is_a ? rows_a : rows_b;
        const int64_t ld = is_a ? p.lda : p.ldb;
        pl.kpos[j] = (uint32_t)(chunk * 16);
        pl.voff[j] = row < rows ? (uint32_t)(row * ld + chunk * 16) : kOOB;
    }

    // ---- fragment read offsets (lane constant): row r = lane & 15, lane group g = lane >> 4
    //      reads chunk g and chunk 4 + g of its row, swizzled by (r >> 1) -----
    const int fr = lane & 15, fg = lane >> 4;
    const uint32_t off1 = (uint32_t)(fr * BK + ((fg ^ (fr >> 1)) << 4));
    const uint32_t off2 = (uint32_t)(fr * BK + (((4 + fg) ^ (fr >> 1)) << 4));

    f32x4 acc[C::TN][C::TM];
    run_tile<C, false>(p, smem, pl, ra, rb, wave, wm0, wn0, off1, off2, acc);

    if (p.nan_zero) {
        int bad = 0;
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
                for (int j = 0; j < 4; ++j) bad |= (acc[tn][tm][j] != acc[tn][tm][j]) ? 1 : 0;
        // block-wide OR through the (now idle) staging buffer: no second LDS object
        volatile int *flag = (volatile int *)smem;
        if (threadIdx.x == 0) *flag = 0;
        __syncthreads();
        if (bad) *flag = 1;
        __syncthreads();
        const int any_bad = *flag;
        __syncthreads();
        if (any_bad) run_tile<C, true>(p, smem, pl, ra, rb, wave, wm0, wn0, off1, off2, acc);
    }

    // ---- fused epilogue ---------------------------------------------------
    const bool has_bias = p.bias != nullptr, has_sr = p.scale_result != nullptr;
    const float sr = has_sr ? p.scale_result[0] : 1.0f;
    const float sa0 = p.scale_a[0], sb0 = p.scale_b[0];
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm) {
        const int64_t m = m0 + wm0 + tm * 16 + fr;
        if (m >= p.M) continue;
        const float sa = p.sa_row ? p.scale_a[m] : sa0;
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn) {
            const int64_t n = n0 + wn0 + tn * 16 + fg * 4;
            if (n >= p.N) continue;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t nj = min(n + j, p.N - 1);
                const float sb = p.sb_row ? p.scale_b[nj] : sb0;
                const float b = has_bias ? load_as_float(p.bias, nj, p.bias_dtype) : 0.0f;
                v[j] = epilogue_value(acc[tn][tm][j], sa, sb, has_bias, b, has_sr, sr);
            }
            const int64_t idx = m * p.ldc + n;
            if (vec_store && n + 3 < p.N) {
                if (p.out_dtype == FP8MI_F32) {
                    *(f32x4 *)((float *)p.C + idx) = f32x4{v[0], v[1], v[2], v[3]};
                } else if (p.out_dtype == FP8MI_BF16) {
                    __bf16 h0 = (__bf16)v[0], h1 = (__bf16)v[1], h2 = (__bf16)v[2], h3 = (__bf16)v[3];
                    u32x2 pk = {(uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16),
                                (uint32_t)__builtin_bit_cast(uint16_t, h2) | ((uint32_t)__builtin_bit_cast(uint16_t, h3) << 16)};
                    *(u32x2 *)((uint16_t *)p.C + idx) = pk;
                } else {
                    _Float16 h0 = (_Float16)v[0], h1 = (_Float16)v[1], h2 = (_Float16)v[2], h3 = (_Float16)v[3];
                    u32x2 pk = {(uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16),
                                (uint32_t)__builtin_bit_cast(uint16_t, h2) | ((uint32_t)__builtin_bit_cast(uint16_t, h3) << 16)};
                    *(u32x2 *)((uint16_t *)p.C + idx) = pk;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (n + j < p.N) store_from_float(p.C, idx + j, v[j], p.out_dtype);
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int NSTAGE>
int launch(const MMParams &p, hipStream_t s)
{
    using C = Cfg<BM, BN, WM, WN, NSTAGE>;
    const int64_t tm = (p.M + BM - 1) / BM, tn = (p.N + BN - 1) / BN;
    if (tm * tn > 0x7FFFFFFF) return FP8MI_E_UNSUPPORTED;
    const int esz = p.out_dtype == FP8MI_F32 ? 4 : 2;
    const int vec = ((p.ldc % 4) == 0 && (((uintptr_t)p.C) % (4 * esz)) == 0) ? 1 : 0;
    FP8MI_LAUNCH((gemm_kernel<BM, BN, WM, WN, NSTAGE>), dim3((unsigned)(tm * tn)), dim3(C::kThreads), s, p, (int)tm,
                       vec);
    return (int)hipGetLastError();
}

}  // namespace

bool fp8mi_gemm_supported(const MMParams &p)
{
    return p.M >= 1 && p.N >= 1 && p.K >= 0 && (p.K % 16) == 0 && (p.lda % 16) == 0 && (p.ldb % 16) == 0 &&
           (((uintptr_t)p.A) & 15u) == 0 && (((uintptr_t)p.B) & 15u) == 0 && p.lda < (1 << 22) && p.ldb < (1 << 22);
}

int fp8mi_launch_gemm(const MMParams &p, int variant, hipStream_t s)
{
    if (variant == FP8MI_KERNEL_AUTO) {
        const int64_t t256 = ((p.M + 255) / 256) * ((p.N + 255) / 256);
        const int64_t t128 = ((p.M + 127) / 128) * ((p.N + 127) / 128);
        if (t256 >= 192) variant = FP8MI_KERNEL_GEMM_256;
        else if (t128 >= 192) variant = FP8MI_KERNEL_GEMM_128;
        else variant = FP8MI_KERNEL_GEMM_128x64;
    }
    switch (variant) {
    case FP8MI_KERNEL_GEMM_128: return launch<128, 128, 64, 64, 4>(p, s);     // 4 x 32 KiB ring
    case FP8MI_KERNEL_GEMM_128x64: return launch<128, 64, 64, 32, 6>(p, s);    // 6 x 24 KiB ring
    case FP8MI_KERNEL_GEMM_256: return launch<256, 256, 128, 64, 2>(p, s);     // 2 x 64 KiB
    default: return FP8MI_E_ENUM;
    }
}
