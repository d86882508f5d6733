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
//     profiles/r01_v1_*).  One barrier per K-step.  (Ping-pong wave groups, an
//     in-wave software pipeline and interleaved DMA issue were built and
//     measured too - none beat this loop, see DESIGN.md 6 - and were removed.)
//   * output orientation: the W fragment is the MFMA "A" operand and the X
//     fragment the "B" operand, so each lane ends up with 4 CONSECUTIVE n of
//     one row m in an accumulator register quad -> one 16-byte store.
//   * the epilogue (scales, bias, result scale, cast) is fused.
//   * block -> tile mapping is XCD-aware: the 8 XCDs get contiguous runs of
//     tiles in a grouped order (4 m-tiles x all n-tiles per group), so the
//     tiles an XCD runs at one time share a few A and B panels in its L2.
//   * split-K (fp8mi_scaled_mm_ws): when the tile grid leaves most CUs idle,
//     K is cut into slices, one workgroup per (tile, slice); fp32 partial
//     tiles meet in a caller-owned workspace and the last workgroup of a tile
//     to arrive adds them in slice order and runs the epilogue.
//
// NaN bytes: the hardware treats 0x7F/0xFF as NaN; the reference decodes them
// to 0.0 (fp8_matmul.metal:21).  The K loop runs unscrubbed; because finite
// e4m3 products cannot overflow fp32, a NaN accumulator proves a NaN byte was
// involved, and only then the workgroup redoes its tile with a SWAR scrub of
// every fragment.  Clean inputs (everything the reference's encoder can emit)
// never pay for it.

#include "fp8mi_gemm_epi.h"
#include "fp8mi_dispatch.h"

namespace {

template <int BM, int BN, int WM, int WN, int NSTAGE_, int MODE_ = 0, int ABL_ = 0, int KS_ = 1, int LD_ = 0>
struct Cfg {
    static constexpr int KS = KS_;      // K-steps (of 128 bytes) per ring stage: KS = 2 halves the barriers per byte
    static constexpr int PFD = ABL_ & 7;  // L2 prefetch distance in ring stages beyond the stage being staged (0 = none): see prefetch_stage
    static constexpr int PFA = (ABL_ >> 3) & 1;  // diagnostic variants: one more wave warms this tile's share of its A panel's lines too
    // Timing-only FLOOR forms of a tile kernel (wrong results; instantiated ONLY by tools/ceiling_probe.hip, which compiles this file with
    // FP8MI_FLOOR_PROBE for bench.py's `roofline.floor`): 1 = the K loop's LDS-DMA stream, waits and barriers without fragment reads and MFMAs;
    // 2 = no K loop at all (launch, arguments, tile map, the C store of the fused epilogue, kernel end); 3 = return behind the argument loads
    // (the launch itself with this kernel's grid, block and LDS allocation).  What each leaves out is hidden under the others in the real
    // kernel, so the floors bound the kernel from below; they do not add up to it.
    static constexpr int FLOOR = (ABL_ >> 4) & 3;
    // Timing-only, diagnostic library (ids 137, 138): the B operand's stage loads go into a register sink as plain `buffer_load_dwordx4` instead of
    // through the LDS-DMA (the B half of the ring stays zero: wrong results).  An upper bound on what a register-staged W loader could gain on the
    // weight-streaming shapes, where plain loads were probed 10-15 % faster than the DMA path (profiles/r04_wstream_probe.txt).
    static constexpr bool BREG = ((ABL_ >> 6) & 1) != 0;
    // Timing experiment, diagnostic library (id 144): the K slices of a tile on ONE XCD (consecutive entries of the XCD's run of the work list) and the
    // split-K exchange at GROUP scope - stores that stop at the XCD's L2, loads that start there, the arrival counter in it.  Right only as long as
    // workgroup b really runs on XCD b mod 8, which the product never relies on for correctness (fp8mi_gemm_epi.h): it measures what that reliance would buy.
    static constexpr bool XLOCAL = ((ABL_ >> 7) & 1) != 0;
    static constexpr int MODE = MODE_;  // 0: stage DMA issued first, then fragment reads + MFMAs; 1: fragment reads first (see run_tile)
    static constexpr int NSTAGE = NSTAGE_;
    static constexpr int PF = NSTAGE_ - 1;  // K-steps of loads in flight
    static constexpr int kWavesM = BM / WM;
    static constexpr int kWavesN = BN / WN;
    static constexpr int kWaves = kWavesM * kWavesN;
    static constexpr int kThreads = kWaves * 64;
    static constexpr int kCThreads = kThreads;  // every wave holds accumulators
    static constexpr int TM = WM / 16;
    static constexpr int TN = WN / 16;
    static constexpr int kGroupsA = BM / 8;  // 8-row staging groups
    static constexpr int kGroupsB = BN / 8;
    static constexpr int kGroups = kGroupsA + kGroupsB;         // per K-step
    static constexpr int kStepBytes = (BM + BN) * BK;
    // Waves that issue the LDS-DMA (0 = all).  Every 1-KiB DMA instruction takes ~16 cycles in the CU's
    // one address path and blocks its wave until accepted; with all waves loading, both waves of a SIMD
    // sit in that queue at the head of each K-step and the matrix pipe idles (measured with in-kernel
    // stamps).  With kLoaders = kWaves / 2 only waves 0..kLoaders-1 (one per SIMD) load, and their SIMD
    // partners start the step's fragment reads and MFMAs at once.
    static constexpr int kLoaders = LD_ == 0 ? kWaves : LD_;
    static constexpr int kGroupsPerWave = KS_ * kGroups / kLoaders;  // per stage, per loading wave
    static constexpr int kStageBytes = KS_ * kStepBytes;
    static_assert(kGroupsA % kLoaders == 0 && kGroupsB % kLoaders == 0 && kLoaders % 2 == 0 && kLoaders <= kWaves,
                  "each loading wave stages whole groups of both operands; the shared swizzle needs an even count");
    static constexpr int kRingBytes = NSTAGE_ * kStageBytes;
    static_assert(NSTAGE_ >= 2 && NSTAGE_ <= 6 && (NSTAGE_ - 1) * kGroupsPerWave <= 63, "vmcnt is a 6-bit counter");
    static_assert(kRingBytes + 16 <= 160 * 1024, "LDS is 160 KiB per CU");
    static_assert(MODE_ >= 0 && MODE_ <= 2 && (MODE_ != 2 || KS_ == 1), "three orders of the loop body are built (the staggered one for KS = 1)");
};

// one K-step's fragments: LDS -> registers
template <typename C, bool SCRUB>
FP8MI_DEVICE void load_frags(const uint8_t *a_rows, const uint8_t *b_rows, int a_row0 /* m */, int b_row0 /* n */,
                             uint32_t off1, uint32_t off2, i32x8 (&xf)[C::TM], i32x8 (&wf)[C::TN])
{
    const uint8_t *sa = a_rows + a_row0 * BK;   // X rows (m)
    const uint8_t *sB = b_rows + b_row0 * BK;   // W rows (n)
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
}

// registers -> MFMAs (W fragment is the MFMA "A" operand, X fragment the "B" operand)
template <typename C>
FP8MI_DEVICE void mfma_all(const i32x8 (&xf)[C::TM], const i32x8 (&wf)[C::TN], f32x4 (&acc)[C::TN][C::TM])
{
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm)
            acc[tn][tm] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[tn], xf[tm], acc[tn][tm], 0, 0, 0,
                                                                            kScaleOne, 0, kScaleOne);
}

template <typename C, bool SCRUB>
FP8MI_DEVICE void compute_step(const uint8_t *stage, int a_row0, int b_row0, uint32_t off1, uint32_t off2,
                               f32x4 (&acc)[C::TN][C::TM])
{
    i32x8 xf[C::TM], wf[C::TN];
    load_frags<C, SCRUB>(stage, stage + C::kGroupsA * 1024, a_row0, b_row0, off1, off2, xf, wf);  // per K-step: A's rows, then B's
    mfma_all<C>(xf, wf, acc);
}

// Per-lane staging plan.  A wave stages kGroupsPerWave 1-KiB groups per stage
// (8 rows x 128 B each).  Its groups of one operand are kWaves * 8 rows apart and
// - because kWaves is even - share one swizzled chunk, so the whole plan is two
// base offsets, a row and a K position; rows are only bounds-checked in ragged
// (edge) tiles.
template <typename C>
struct StagePlan {
    uint32_t va0, vb0;  // byte offset of this lane's 16 bytes in the wave's first A / B group
    uint32_t row0;      // its row there (the same for A and B)
    uint32_t kpos;      // chunk * 16: position inside the 128-byte K-step
    uint32_t sa, sb;    // uniform byte stride between the wave's consecutive A / B groups
    int rows_a, rows_b; // valid rows of this tile
    bool full;          // interior tile: no row masking
    uint32_t pf_off;    // C::PFD: this lane's line of the B panel's share to prefetch (kOOB: none)
    u32x4 pf_rsrc;      // ... and the B panel's descriptor as plain words (an asm operand)
    int pf_waves;       // how many waves prefetch (1 when the panel has 4 readers on the XCD, 2 with 2, 0 when this tile is its only reader)
    mutable u32x4 bsink;  // C::BREG (timing-only): landing registers of the B operand's plain loads (never read)
};

template <typename C, bool TAIL>
FP8MI_DEVICE void issue_stage(const StagePlan<C> &pl, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb,
                              uint8_t *stage, int wave, int k0, int64_t K)
{
    constexpr int JA = C::kGroupsA / C::kLoaders, JB = C::kGroupsB / C::kLoaders, JS = JA + JB;
    if (C::kLoaders < C::kWaves && wave >= C::kLoaders) return;  // wave-uniform: this wave does not load
#pragma unroll
    for (int j = 0; j < C::kGroupsPerWave; ++j) {
        const int q = j / JS, jj = j % JS;            // K-step inside the stage, group inside the K-step
        const bool is_a = jj < JA;
        const int jo = is_a ? jj : jj - JA;
        const int gs = q * C::kGroups + (is_a ? 0 : C::kGroupsA) + wave + jo * C::kLoaders;  // LDS group slot (wave-uniform)
        uint32_t vo = (is_a ? pl.va0 + jo * pl.sa : pl.vb0 + jo * pl.sb) + q * BK;
        if (!pl.full && (int)(pl.row0 + jo * C::kLoaders * 8) >= (is_a ? pl.rows_a : pl.rows_b)) vo = kOOB;
        if (TAIL && (int64_t)k0 + q * BK + pl.kpos >= K) vo = kOOB;  // K tail: only in the peeled last step
        // the LDS image is consecutive 1-KiB groups (8 rows x 128 B), per K-step A's rows first, then B's
        lds_void *dst = (lds_void *)(stage + gs * 1024);
        if (is_a) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, dst, 16, (int)vo, k0, 0, 0);
        else if constexpr (C::BREG) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "+v"(pl.bsink) : "v"(vo), "s"(pl.pf_rsrc), "s"((uint32_t)k0) : "memory");
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, dst, 16, (int)vo, k0, 0, 0);
    }
}

// Diagnostic build only (-DFP8MI_STAMP, never shipped): per-phase cycle sums of
// wave 0 of every workgroup, written over the first bytes of each C tile row 0.
#ifdef FP8MI_STAMP
#define STAMP(var)                                                                 \
    do {                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                         \
        unsigned long long t_;                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                         \
        var = t_;                                                                  \
    } while (0)
__device__ unsigned long long g_stamp[256 * 32];  // per block: [wave 0 | wave kWaves/2] x 8 sums, then their 5 absolute stamps of step 5
#else
#define STAMP(var) do { } while (0)
#endif

template <int N>
FP8MI_DEVICE void wait_loads_and_lds()
{
    // this wave's LDS-DMA except the N youngest have landed; its ds_reads are done
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

// wait until at most `newer_stages` stages' worth of this wave's LDS-DMA is outstanding
// (loads retire in issue order, so the older stage has landed), and its ds_reads are done
template <typename C>
FP8MI_DEVICE void wait_stage(int newer_stages)
{
    constexpr int G = C::kGroupsPerWave;
    if (5 * G <= 63 && newer_stages >= 5) wait_loads_and_lds<(5 * G <= 63 ? 5 * G : 0)>();
    else if (4 * G <= 63 && newer_stages == 4) wait_loads_and_lds<(4 * G <= 63 ? 4 * G : 0)>();
    else if (3 * G <= 63 && newer_stages == 3) wait_loads_and_lds<(3 * G <= 63 ? 3 * G : 0)>();
    else if (2 * G <= 63 && newer_stages == 2) wait_loads_and_lds<(2 * G <= 63 ? 2 * G : 0)>();
    else if (newer_stages == 1) wait_loads_and_lds<G>();
    else wait_loads_and_lds<0>();
}

template <typename C>
FP8MI_DEVICE void issue_any(const StagePlan<C> &pl, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb,
                            uint8_t *stage, int wave, int step, int nk, bool ktail, int64_t K)
{
    if (ktail && step == nk - 1) issue_stage<C, true>(pl, ra, rb, stage, wave, step * (BK * C::KS), K);
    else issue_stage<C, false>(pl, ra, rb, stage, wave, step * (BK * C::KS), K);
}

// L2 prefetch (C::PFD > 0).  The tiles an XCD runs at one time are 4 m-tiles x 8 n-tiles (tile_of_block): a B panel - the
// weights, streamed from HBM - has 4 readers that ask for the same lines at the same time, and every one of them waits out
// the HBM latency inside the ring's prefetch window.  Wave 0 of each tile touches one dword per 128-byte line of ITS quarter of
// the panel for a stage PFD stages beyond the one being staged; the others then hit the L2.  The load's result is never used;
// the register it lands in (`sink`) is carried through the loop so that hipcc does not hand it to anything else.
FP8MI_DEVICE void prefetch_stage(uint32_t &sink, uint32_t voff, u32x4 rsrc, uint32_t koff)
{
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "+v"(sink) : "v"(voff), "s"(rsrc), "s"(koff) : "memory");
}

template <typename C, bool SCRUB>
FP8MI_DEVICE void run_tile(const MMParams &p, uint8_t *smem, const StagePlan<C> &pl, __amdgpu_buffer_rsrc_t ra,
                           __amdgpu_buffer_rsrc_t rb, int wave, int wm0, int wn0, uint32_t off1, uint32_t off2,
                           int ks0, int nk, int rot, f32x4 (&acc)[C::TN][C::TM])
{
    // ks0, nk: this workgroup's range of ring stages (all of K, or one split-K slice); rot in [0, nk)
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm) acc[tn][tm] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    const int64_t K = p.K;
    const int nk_all = (int)((K + BK * C::KS - 1) / (BK * C::KS));
    const bool ktail = (K % (BK * C::KS)) != 0;  // then the tile's last stage is staged with per-lane K masking
    if (nk == 0) return;
    unsigned long long p0_ = 0, p1_ = 0, p2_ = 0; (void)p0_; (void)p1_; (void)p2_;
    STAMP(p0_);

    // The K loop is walked circularly from `rot` (a per-m-tile offset): the tiles
    // that share a B panel run at the same time on one XCD, and if they all
    // started at k = 0 each of them would wait out HBM latency on the same lines;
    // staggered, each one fetches a different K range from HBM and finds the
    // rest already in the XCD's L2.  (Only the summation order changes.)
    int ks = rot;  // stage (relative to ks0) the next issue loads
    auto next_ks = [&]() { const int r = ks; ks = (ks + 1 == nk) ? 0 : ks + 1; return ks0 + r; };
    uint32_t sink = 0;  // C::PFD: landing register of the L2 prefetch loads (never read)
    auto prefetch = [&](int stage) {  // stage relative to ks0, clamped to this workgroup's last one
        if (C::PFD > 0 && wave < pl.pf_waves + C::PFA) prefetch_stage(sink, pl.pf_off, pl.pf_rsrc, (uint32_t)((ks0 + min(stage, nk - 1)) * (BK * C::KS)));
    };

    // prologue: PF stages in flight (stage s -> ring slot s)
#pragma unroll
    for (int s = 0; s < C::PF; ++s)
        if (s < nk) issue_any<C>(pl, ra, rb, smem + s * C::kStageBytes, wave, next_ks(), nk_all, ktail, K);
#pragma unroll
    for (int s = 0; s < C::PFD; ++s) prefetch(C::PF + s);

    STAMP(p1_);
    int slot = 0;             // ring slot of step t
    int fill = C::PF % C::NSTAGE;  // ring slot the next issue goes to (= slot of step t-1)
    unsigned long long c_wait = 0, c_bar = 0, c_issue = 0, c_comp = 0, s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
    (void)c_wait; (void)c_bar; (void)c_issue; (void)c_comp; (void)s0; (void)s1; (void)s2; (void)s3; (void)s4;
    for (int t = 0; t < nk; ++t) {
        STAMP(s0);
        // stage t has landed for this wave once at most the newer stages' loads are outstanding
        wait_stage<C>(min(C::PF - 1, nk - 1 - t));
        STAMP(s1);
        __builtin_amdgcn_s_barrier();  // ... for every wave; and every wave is done reading slot of step t-1
        STAMP(s2);
        if constexpr (C::MODE == 1) {
            // reads first: this K-step's fragment reads are issued BEFORE the stage DMA, so that they land while the
            // loading waves sit in the DMA issue (a `buffer_load ... lds` holds its wave ~64 cycles: 1,085 cycles per
            // K-step on the 256x256 tile, in-kernel stamps) and the MFMAs start the moment the issue is through
            i32x8 xf[C::TM], wf[C::TN];
            const uint8_t *st = smem + slot * C::kStageBytes;
            if constexpr (C::FLOOR == 0) load_frags<C, SCRUB>(st, st + C::kGroupsA * 1024, wm0, wn0, off1, off2, xf, wf);
            __builtin_amdgcn_sched_barrier(0);
            if (t + C::PF < nk) {
                issue_any<C>(pl, ra, rb, smem + fill * C::kStageBytes, wave, next_ks(), nk_all, ktail, K);
                prefetch(t + C::PF + C::PFD);
            }
            STAMP(s3);
            if constexpr (C::FLOOR == 0) {
                mfma_all<C>(xf, wf, acc);
#pragma unroll
                for (int q = 1; q < C::KS; ++q)
                    compute_step<C, SCRUB>(st + q * C::kStepBytes, wm0, wn0, off1, off2, acc);
            }
        } else {
            if (t + C::PF < nk) {
                issue_any<C>(pl, ra, rb, smem + fill * C::kStageBytes, wave, next_ks(), nk_all, ktail, K);
                prefetch(t + C::PF + C::PFD);
            }
            STAMP(s3);
            if constexpr (C::FLOOR == 0) {
#pragma unroll
                for (int q = 0; q < C::KS; ++q)
                    compute_step<C, SCRUB>(smem + slot * C::kStageBytes + q * C::kStepBytes, wm0, wn0, off1, off2, acc);
            }
        }
        STAMP(s4);
        c_wait += s1 - s0; c_bar += s2 - s1; c_issue += s3 - s2; c_comp += s4 - s3;
#ifdef FP8MI_STAMP
        if (t == 5 && (threadIdx.x & 63) == 0 && blockIdx.x < 256 && (wave == 0 || wave == C::kWaves / 2)) {
            unsigned long long *o = g_stamp + blockIdx.x * 32 + 16 + (wave ? 5 : 0);
            o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3; o[4] = s4;
        }
#endif
        slot = (slot + 1 == C::NSTAGE) ? 0 : slot + 1;
        fill = (fill + 1 == C::NSTAGE) ? 0 : fill + 1;
    }
#ifdef FP8MI_STAMP
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 256 && (wave == 0 || wave == C::kWaves / 2)) {
        unsigned long long *o = g_stamp + blockIdx.x * 32 + (wave ? 8 : 0);
        o[0] = c_wait; o[1] = c_bar; o[2] = c_issue; o[3] = c_comp; o[4] = nk;
    }
#endif
    // all loads were waited for in the last iteration (newer_stages == 0); the caller's barrier makes the ring reusable
    if (C::PFD > 0) asm volatile("" ::"v"(sink));  // the prefetch loads' landing register stays reserved up to here
#ifdef FP8MI_STAMP
    STAMP(p2_);
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 256 && wave == 0) {
        unsigned long long *o = g_stamp + blockIdx.x * 32 + 26;
        o[0] = p0_; o[1] = p1_; o[2] = p2_;
    }
#endif
}

// STAGGERED wave groups (MODE 2).  Left alone all 8 waves run in lockstep: first everybody reads fragments (the LDS is
// the bottleneck and the matrix pipes idle: 192 KiB per K-step of a 256x256 tile = 768 cycles at 256 B/clk), then everybody
// multiplies (the LDS idles).  Here waves kWaves/2.. ("late") run half a step behind their SIMD partners: they multiply
// K-step t - 1, whose fragments they kept in registers across the barrier, while waves 0..kWaves/2-1 read K-step t; then
// they read K-step t while the partners multiply it.  Same instructions per wave, same registers, same number of
// barriers, bit-identical results; two separate loops so that each keeps one clean register assignment.
template <typename C, bool SCRUB>
FP8MI_DEVICE void run_tile_staggered(const MMParams &p, uint8_t *smem, const StagePlan<C> &pl, __amdgpu_buffer_rsrc_t ra,
                                     __amdgpu_buffer_rsrc_t rb, int wave, int wm0, int wn0, uint32_t off1, uint32_t off2,
                                     int ks0, int nk, int rot, f32x4 (&acc)[C::TN][C::TM])
{
    static_assert(C::KS == 1, "one K-step per ring stage");
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm) acc[tn][tm] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int64_t K = p.K;
    const int nk_all = (int)((K + BK - 1) / BK);
    const bool ktail = (K % BK) != 0;
    if (nk == 0) return;
    int ks = rot;
    auto next_ks = [&]() { const int r = ks; ks = (ks + 1 == nk) ? 0 : ks + 1; return ks0 + r; };
#pragma unroll
    for (int s = 0; s < C::PF; ++s)
        if (s < nk) issue_any<C>(pl, ra, rb, smem + s * C::kStageBytes, wave, next_ks(), nk_all, ktail, K);
    int slot = 0, fill = C::PF % C::NSTAGE;
    auto advance = [&]() {
        slot = (slot + 1 == C::NSTAGE) ? 0 : slot + 1;
        fill = (fill + 1 == C::NSTAGE) ? 0 : fill + 1;
    };
    i32x8 xf[C::TM], wf[C::TN];
    if (wave < C::kWaves / 2) {  // early group: read, (load,) multiply
        for (int t = 0; t < nk; ++t) {
            wait_stage<C>(min(C::PF - 1, nk - 1 - t));
            __builtin_amdgcn_s_barrier();
            const uint8_t *st = smem + slot * C::kStageBytes;
            load_frags<C, SCRUB>(st, st + C::kGroupsA * 1024, wm0, wn0, off1, off2, xf, wf);
            __builtin_amdgcn_sched_barrier(0);
            if (t + C::PF < nk) issue_any<C>(pl, ra, rb, smem + fill * C::kStageBytes, wave, next_ks(), nk_all, ktail, K);
            mfma_all<C>(xf, wf, acc);
            advance();
        }
    } else {                     // late group: multiply the previous K-step, (load,) read this one
        for (int t = 0; t < nk; ++t) {
            wait_stage<C>(min(C::PF - 1, nk - 1 - t));
            __builtin_amdgcn_s_barrier();
            if (t > 0) {
                mfma_all<C>(xf, wf, acc);
#pragma unroll
                for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
                    for (int tm = 0; tm < C::TM; ++tm) asm volatile("" : "+v"(acc[tn][tm]));  // the MFMAs stay above the reads
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (t + C::PF < nk) issue_any<C>(pl, ra, rb, smem + fill * C::kStageBytes, wave, next_ks(), nk_all, ktail, K);
            const uint8_t *st = smem + slot * C::kStageBytes;
            load_frags<C, SCRUB>(st, st + C::kGroupsA * 1024, wm0, wn0, off1, off2, xf, wf);
            advance();
        }
        mfma_all<C>(xf, wf, acc);  // the last K-step
    }
}

template <typename C, bool SCRUB>
FP8MI_DEVICE void run_tile_any(const MMParams &p, uint8_t *smem, const StagePlan<C> &pl, __amdgpu_buffer_rsrc_t ra,
                               __amdgpu_buffer_rsrc_t rb, int wave, int wm0, int wn0, uint32_t off1, uint32_t off2,
                               int ks0, int nk, int rot, f32x4 (&acc)[C::TN][C::TM])
{
    if constexpr (C::MODE == 2) run_tile_staggered<C, SCRUB>(p, smem, pl, ra, rb, wave, wm0, wn0, off1, off2, ks0, nk, rot, acc);
    else run_tile<C, SCRUB>(p, smem, pl, ra, rb, wave, wm0, wn0, off1, off2, ks0, nk, rot, acc);
}

template <int BM, int BN, int WM, int WN, int NSTAGE, int PP, int ABL, int KS, int LD>
__global__ __launch_bounds__((Cfg<BM, BN, WM, WN, NSTAGE, PP, ABL, KS, LD>::kThreads)) void gemm_kernel(MMParams p_in, int tiles_m, int tiles_n, int vec_store, int nwg)
{
    using C = Cfg<BM, BN, WM, WN, NSTAGE, PP, ABL, KS, LD>;
    const MMParams p = pin_params(p_in);  // every kernel argument in one scalar-load clause (fp8mi_common.h)
    FP8MI_PIN_S(tiles_m); FP8MI_PIN_S(tiles_n); FP8MI_PIN_S(vec_store); FP8MI_PIN_S(nwg);
    if constexpr (C::FLOOR == 3) {   // timing-only (tools/ceiling_probe.hip): the launch alone - arguments fetched and used, nothing else
        if (p.M < 0) *(volatile int *)p.C = tiles_m + tiles_n + vec_store + nwg;   // (never true: keeps the argument loads alive)
        return;
    }
    const EpiScalars es = load_epi_scalars(p);  // in flight under the K loop
    __shared__ __attribute__((aligned(16))) uint8_t smem[C::kRingBytes + kFlagBytes];
    if (threadIdx.x == 0) *(volatile int *)(smem + C::kRingBytes) = 0;  // NaN verdict word (ordered by the K loop's barriers)

    unsigned long long k0_ = 0, k1_ = 0, k2_ = 0; (void)k0_; (void)k1_; (void)k2_;
    STAMP(k0_);
#ifdef FP8MI_STAMP
    const unsigned long long r0_ = __builtin_amdgcn_s_memrealtime();
#endif
    int tile_m, tile_n, kslice, wg;
    tile_of_block(blockIdx.x, nwg, tiles_m, tiles_n, tile_m, tile_n, kslice, wg);  // XCD-aware, grouped order (fp8mi_gemm_epi.h)
    if constexpr (C::XLOCAL) {   // (timing experiment) slice fastest inside an XCD's run: a tile's slices sit on one XCD when the run is a multiple of the split
        const int q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7;
        const int wg_all = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + ((int)blockIdx.x >> 3);
        const int split = max(p.split, 1);
        wg = wg_all / split;
        kslice = wg_all - wg * split;
        tile_m = wg % tiles_m;
        tile_n = wg / tiles_m;
    }
    const int n_tiles = tiles_m * tiles_n;
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
    {
        const int row0 = wave * 8 + (lane >> 3);                           // row of this lane in the wave's first group
        const int chunk = (lane & 7) ^ (((wave & 1) * 4 + (lane >> 4)) & 7);  // = (lane & 7) ^ ((row >> 1) & 7) for every group
        pl.row0 = (uint32_t)row0;
        pl.kpos = (uint32_t)(chunk * 16);
        pl.va0 = (uint32_t)(row0 * p.lda + chunk * 16);
        pl.vb0 = (uint32_t)(row0 * p.ldb + chunk * 16);
        pl.sa = (uint32_t)(C::kLoaders * 8 * p.lda);
        pl.sb = (uint32_t)(C::kLoaders * 8 * p.ldb);
        pl.rows_a = (int)rows_a;
        pl.rows_b = (int)rows_b;
        pl.full = rows_a == BM && rows_b == BN;
        pl.pf_off = kOOB;
        pl.pf_waves = 0;
        pl.bsink = u32x4{0u, 0u, 0u, 0u};
        if constexpr (C::BREG) {   // (timing-only) the B panel's descriptor as plain words for the asm loads; the ring starts out zero: no NaN redo on stale LDS bytes
            static_assert(C::PFD == 0, "BREG borrows the prefetch descriptor");
            const uint64_t pbr = (uint64_t)(p.B + n0 * p.ldb);
            pl.pf_rsrc = u32x4{(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pbr),
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(pbr >> 32) & 0xFFFFu)),
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)min(bytes_b, (int64_t)0x7FFFFFFF)), 0x00020000u};
            for (int o = (int)threadIdx.x * 16; o < C::kRingBytes; o += C::kThreads * 16) *(u32x4 *)(smem + o) = u32x4{0u, 0u, 0u, 0u};
            __syncthreads();
        }
        if (C::PFD > 0) {
            // this tile's quarter of its B panel's lines for one stage (B has 4 readers on the XCD: the m-tiles of its group)
            constexpr int kLines = BN * C::KS / 4;
            static_assert(C::PFD == 0 || kLines <= 64, "one prefetch instruction per stage");
            // readers of this B panel on the XCD = the m-tiles of this tile's group (4, fewer in a short last group): with 4 readers
            // wave 0 of each warms one quarter, with 2 readers waves 0-1 of each.  A tile that is its panel's only reader does not
            // prefetch: warming its own lines only adds requests (decode shape M=64 K=14336 N=4096: 18.6 -> 20.5 us, measured)
            const int gm = min(4, tiles_m - (tile_m & ~3)), nshare = gm >= 4 ? 4 : (gm >= 2 ? 2 : 1);
            pl.pf_waves = nshare == 1 ? 0 : 4 / nshare;
            const int quarter = (tile_m % nshare) * (4 / nshare) + wave;
            const int li = (quarter & 3) * kLines + lane, prow = li / C::KS, pk = li % C::KS;
            if (lane < kLines && prow < (int)rows_b && (p.K % (BK * C::KS)) == 0) pl.pf_off = (uint32_t)(prow * p.ldb + pk * BK);
            const uint64_t pb = (uint64_t)(p.B + n0 * p.ldb);
            pl.pf_rsrc = u32x4{(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pb),
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(pb >> 32) & 0xFFFFu)),
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)min(bytes_b, (int64_t)0x7FFFFFFF)), 0x00020000u};
            if (C::PFA && wave == pl.pf_waves) {
                // the wave behind the B-prefetching ones takes this tile's eighth of its A panel's lines (the 8 n-tiles an XCD runs
                // at one time share the panel; A comes from the Infinity Cache for all but the first XCD to touch it)
                constexpr int kLinesA = BM * C::KS / 8;
                static_assert(!C::PFA || kLinesA <= 64, "one prefetch instruction per stage");
                const int lia = (tile_n & 7) * kLinesA + lane, arow = lia / C::KS, ak = lia % C::KS;
                pl.pf_off = (lane < kLinesA && arow < (int)rows_a && (p.K % (BK * C::KS)) == 0) ? (uint32_t)(arow * p.lda + ak * BK) : kOOB;
                const uint64_t pa = (uint64_t)(p.A + m0 * p.lda);
                pl.pf_rsrc = u32x4{(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pa),
                                   (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(pa >> 32) & 0xFFFFu)),
                                   (uint32_t)__builtin_amdgcn_readfirstlane((int)min(bytes_a, (int64_t)0x7FFFFFFF)), 0x00020000u};
            }
        }
    }

    // ---- fragment read offsets (lane constant): row r = lane & 15, lane group g = lane >> 4
    //      reads chunk g and chunk 4 + g of its row, swizzled by (r >> 1) -----
    const int fr = lane & 15, fg = lane >> 4;
    const uint32_t off1 = (uint32_t)(fr * BK + ((fg ^ (fr >> 1)) << 4));
    const uint32_t off2 = (uint32_t)(fr * BK + (((4 + fg) ^ (fr >> 1)) << 4));

    const int nk_all = (int)((p.K + BK * C::KS - 1) / (BK * C::KS));
    const int nsplit = p.split > 1 ? p.split : 1;
    const int nk_slice = (nk_all + nsplit - 1) / nsplit;           // the host made every slice non-empty
    const int ks0 = kslice * nk_slice, nk = min(nk_slice, nk_all - ks0);
    // K is always walked from 0: every tile kernel then adds the K-steps of an output element in the same order, so the
    // unsplit result does not depend on the tile shape or on where the tile sits (a sharded linear equals the unsharded one
    // bit for bit); a per-m-tile rotated start measured within +-2 % of this
    const int rot = 0;

    f32x4 acc[C::TN][C::TM];
    if constexpr (C::FLOOR == 2) {   // timing-only: no K loop (the accumulators are zero: the epilogue stores a tile of zeros)
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm) acc[tn][tm] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    } else
    run_tile_any<C, false>(p, smem, pl, ra, rb, wave, wm0, wn0, off1, off2, ks0, nk, rot, acc);

    // ---- end of the K loop: one barrier frees the ring and carries the NaN verdict (fp8mi_gemm_epi.h) ----
    volatile int *flag = (volatile int *)(smem + C::kRingBytes);
    if (p.nan_zero && acc_has_nan<C>(acc)) *flag = 1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (p.nan_zero && *flag) {  // workgroup-uniform: redo the tile with every fragment scrubbed (reference NaN semantics)
        run_tile_any<C, true>(p, smem, pl, ra, rb, wave, wm0, wn0, off1, off2, ks0, nk, rot, acc);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- split-K: partial tiles meet in the workspace; only the last-arriving slice runs the epilogue ----
    if (nsplit > 1) {
#ifdef FP8MI_DIAG
        // timing-only bound for a PAIRED exchange: two slices, no exchange at all - each workgroup stores the half of
        // the tile whose rows its slice would own (wrong results: the partner's partial is never added)
        if (p.debug & 1) {
            if ((wave % C::kWavesM) * 2 / C::kWavesM != (kslice & 1)) return;
        } else
#endif
        if (!splitk_combine<C>(p, acc, smem, wg, kslice, nsplit, n_tiles)) return;
    }

    STAMP(k1_);
    // ---- fused epilogue ---------------------------------------------------
    const int rows_m = (int)rows_a, cols_n = (int)rows_b;  // valid extent of this tile
    if (rows_m == BM && cols_n == BN && vec_store) {  // interior tile, 16-byte aligned rows: line-coalesced stores
        if (p.out_dtype == FP8MI_F32) epilogue_staged<C, FP8MI_F32>(p, es, acc, smem, m0, n0, wave, wm0, wn0, lane);
        else if (p.out_dtype == FP8MI_BF16) epilogue_staged<C, FP8MI_BF16>(p, es, acc, smem, m0, n0, wave, wm0, wn0, lane);
        else epilogue_staged<C, FP8MI_F16>(p, es, acc, smem, m0, n0, wave, wm0, wn0, lane);
    } else if (p.out_dtype == FP8MI_F32) epilogue<C, FP8MI_F32>(p, es, acc, m0, n0, wm0, wn0, fr, fg, rows_m, cols_n, vec_store);
    else if (p.out_dtype == FP8MI_BF16) epilogue<C, FP8MI_BF16>(p, es, acc, m0, n0, wm0, wn0, fr, fg, rows_m, cols_n, vec_store);
    else epilogue<C, FP8MI_F16>(p, es, acc, m0, n0, wm0, wn0, fr, fg, rows_m, cols_n, vec_store);
#ifdef FP8MI_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(k2_);
    if (threadIdx.x == 0 && blockIdx.x < 256) {
        g_stamp[blockIdx.x * 32 + 5] = k1_ - k0_;   // entry .. end of K loop (incl. NaN check)
        g_stamp[blockIdx.x * 32 + 6] = k2_ - k1_;   // epilogue incl. store drain
        g_stamp[blockIdx.x * 32 + 7] = __builtin_amdgcn_s_memrealtime() - r0_;  // 100 MHz ticks over the whole tile
        g_stamp[blockIdx.x * 32 + 29] = k0_;
        g_stamp[blockIdx.x * 32 + 30] = k1_;
    }
#endif
}

template <int BM, int BN, int WM, int WN, int NSTAGE, int PP = 0, int ABL = 0, int KS = 1, int LD = 0>
int launch(const MMParams &p_in, hipStream_t s)
{
    using C = Cfg<BM, BN, WM, WN, NSTAGE, PP, ABL, KS, LD>;
    MMParams p = p_in;
    const int64_t tm = (p.M + BM - 1) / BM, tn = (p.N + BN - 1) / BN;
    if (tm * tn > 0x7FFFFFFF) return FP8MI_E_UNSUPPORTED;
    const int rc = resolve_split(p, tm, tn, BM, BN, BK * C::KS);
    if (rc) return rc;
    const int esz = p.out_dtype == FP8MI_F32 ? 4 : 2;
    // 16-byte aligned rows and 4-element groups: enables the vector stores of both epilogues
    const int vec = (((p.ldc * esz) % 16) == 0 && (((uintptr_t)p.C) % 16) == 0) ? 1 : 0;
    const unsigned grid = (unsigned)(tm * tn * p.split);
    return fp8mi_launch(gemm_kernel<BM, BN, WM, WN, NSTAGE, PP, ABL, KS, LD>, dim3(grid), dim3(C::kThreads), s, p, (int)tm, (int)tn, vec,
                        (int)grid);
}

}  // namespace

#ifdef FP8MI_STAMP
extern "C" int fp8mi_debug_read_stamps(unsigned long long *out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * n);
}
#endif

bool fp8mi_gemm_supported(const MMParams &p)
{
    return p.M >= 1 && p.N >= 1 && p.K >= 0 && (p.K % 16) == 0 && (p.lda % 16) == 0 && (p.ldb % 16) == 0 &&
           (((uintptr_t)p.A) & 15u) == 0 && (((uintptr_t)p.B) & 15u) == 0 && p.lda < (1 << 22) && p.ldb < (1 << 22);
}

// The tile kernel the automatic dispatch uses for a shape the tile kernels support: the cheapest tile kernel by the cost model (fp8mi_dispatch.h)
int fp8mi_choose_gemm_variant(const MMParams &p)
{
    const double cus = (double)fp8mi_cu_count();
    int best = FP8MI_KERNEL_GEMM_128x64;
    double best_us = 1e300;
    for (const fp8mi_dispatch::TileCost &t : fp8mi_dispatch::kTileCosts) {
        const double us = fp8mi_dispatch::predict_us(p, t.id, cus);
        if (us >= 0.0 && us < best_us) { best_us = us; best = t.id; }
    }
    return best;
}

int fp8mi_launch_gemm(const MMParams &p, int variant, hipStream_t s)
{
    if (variant == FP8MI_KERNEL_AUTO) variant = fp8mi_choose_gemm_variant(p);
    switch (variant) {
    // product kernels: 8 waves, waves 0-3 (one per SIMD) issue the stage DMA (template: BM, BN, WM, WN, ring stages, loop order,
    // -, K-steps per stage, loading waves).  Loop orders (run_tile / run_tile_staggered): the small tiles issue their fragment reads
    // ahead of the stage DMA (C3 15.9 -> 15.3 us); the 256x256 tile runs its two wave groups half a K-step apart (FLUX 130 -> 124 us,
    // 8192^3 509 -> 479 us); for the 128x128 tile neither order is a consistent gain (shard +1.6 %, 8192^3 -4.5 %): it keeps the plain one.
    case FP8MI_KERNEL_GEMM_128: return launch<128, 128, 64, 32, 2, 0, 0, 1, 4>(p, s);   // 2 x 32 KiB ring: 2 workgroups / CU
    case FP8MI_KERNEL_GEMM_128x64: return launch<128, 64, 32, 32, 3, 1, 1, 2, 4>(p, s);  // 3 x 48 KiB ring, 2 K-steps per stage, L2 prefetch one stage beyond the ring (C3 16.0 -> 15.3 us)
    case FP8MI_KERNEL_GEMM_256: return launch<256, 256, 128, 64, 2, 2, 0, 1, 4>(p, s);   // 2 x 64 KiB ring, staggered wave groups
    case FP8MI_KERNEL_GEMM_64x128: return launch<64, 128, 32, 32, 3, 1, 0, 2, 4>(p, s);  // 3 x 48 KiB ring, for M <= 64
    // small-batch tiles (round 3, tools/sweep_decode.py, profiles/r03_decode_tiles.txt): against a deep K a 64x128 tile x 8 K slices leaves 32 KiB
    // partials and a last arriver that re-reads 256 KiB; 64x64 x 4 slices (16 KiB partials) and, for M <= 32, 32x64 tiles (half the x traffic)
    // took 13-30 % less time on every shape of the sweep (K=14336 N=4096: M=32 17.2 -> 13.2 us, M=64 17.2 -> 14.9 us)
    case FP8MI_KERNEL_GEMM_64x64: return launch<64, 64, 16, 32, 4, 1, 0, 2, 4>(p, s);    // 8 waves of 16x32, 4 x 32 KiB ring, waves 0-3 load
    case FP8MI_KERNEL_GEMM_32x64: return launch<32, 64, 16, 32, 4, 1, 0, 2, 4>(p, s);    // 4 waves of 16x32, 4 x 24 KiB ring
    case FP8MI_KERNEL_GEMM_32x32: return launch<32, 32, 16, 32, 4, 1, 0, 2, 2>(p, s);    // 2 waves of 16x32, 4 x 16 KiB ring: N / 32 tiles need half the K slices (K = N = 8192, M = 32: 14.7 against 18.0 us)
    // One 128x128 tile per CU at most (end of round 3): the 2 x 32 KiB ring above is built for TWO co-resident workgroups, whose other half hides each one's
    // single stage in flight; alone on its CU a workgroup waits out a memory round trip per K-step (0.8 us per step).  The same tile on 4 x 32 KiB (three
    // stages in flight, fragment reads ahead of the stage DMA): M=1024 K=N=4096 21.8 us against 28.1 (128x64, two rounds) / 29.2 (256x128W on half the CUs) /
    // 31.1 (this tile, shallow ring); M=512 K=N=8192 39.5 against 52.0; M=256 K=4096 N=14336 23.1 against 31.7 (profiles/r03_deep_ring.txt)
    case FP8MI_KERNEL_GEMM_128D: return launch<128, 128, 64, 32, 4, 1, 0, 1, 4>(p, s);
    case FP8MI_KERNEL_GEMM_256W: return fp8mi_launch_gemm256(p, 0, s);                   // (only chosen above when fp8mi_gemm256_supported)
    case FP8MI_KERNEL_GEMM_256x128W: return fp8mi_launch_gemm256(p, 1000, s);
#ifdef FP8MI_FLOOR_PROBE  // tools/ceiling_probe.hip only (never in libfp8mi.so): timing-only floors of the kernels bench.py's headline workloads run on
    case 901: return launch<128, 64, 32, 32, 3, 1, 1 | (1 << 4), 2, 4>(p, s);   // FP8MI_KERNEL_GEMM_128x64 (config C3): DMA stream only
    case 902: return launch<128, 64, 32, 32, 3, 1, 1 | (2 << 4), 2, 4>(p, s);   //   ... launch + C store only
    case 903: return launch<128, 64, 32, 32, 3, 1, 1 | (3 << 4), 2, 4>(p, s);   //   ... launch only
    case 911: return launch<128, 128, 64, 32, 4, 1, 0 | (1 << 4), 1, 4>(p, s);  // FP8MI_KERNEL_GEMM_128D (the `wide` class): the same three
    case 912: return launch<128, 128, 64, 32, 4, 1, 0 | (2 << 4), 1, 4>(p, s);
    case 913: return launch<128, 128, 64, 32, 4, 1, 0 | (3 << 4), 1, 4>(p, s);
#endif
#ifdef FP8MI_DIAG  // schedule variants kept for A/B timing (diagnostic library only; same results): tools/ab_kernels.py
    case 30: return launch<256, 256, 128, 64, 2, 1, 0, 1, 4>(p, s);            // 256x256, fragment reads before the stage DMA
    case 31: return launch<128, 128, 64, 32, 2, 1, 0, 1, 4>(p, s);             // 128x128, same
    case 32: return launch<128, 64, 32, 32, 3, 0, 0, 2, 4>(p, s);              // 128x64, stage DMA first (the round-1 order)
    case 33: return launch<256, 256, 128, 64, 2, 1>(p, s);                     // 256x256 reads first, all 8 waves load
    case 34: return launch<256, 256, 128, 64, 2, 0, 0, 1, 4>(p, s);            // 256x256, lockstep (the round-1 order)
    case 35: return launch<256, 256, 128, 64, 2, 2>(p, s);                     // 256x256, staggered, all waves load
    case 36: return launch<128, 128, 64, 32, 2, 2, 0, 1, 4>(p, s);             // 128x128, staggered, waves 0-3 load
    case 37: return launch<128, 128, 64, 32, 2, 2>(p, s);                      // 128x128, staggered, all waves load
    case 38: return launch<128, 64, 32, 32, 3, 1, 0, 2, 4>(p, s);              // 128x64 without the L2 prefetch
    case 39: return launch<128, 64, 32, 32, 3, 1, 2, 2, 4>(p, s);              //   ... two stages
    case 120: return launch<128, 128, 64, 32, 2, 0, 1, 1, 4>(p, s);            // 128x128 + L2 prefetch one stage further
    case 121: return launch<128, 128, 64, 32, 2, 0, 2, 1, 4>(p, s);            //   ... two stages
    case 122: return launch<128, 128, 64, 32, 2, 0, 4, 1, 4>(p, s);            //   ... four stages
    case 130: return launch<128, 128, 64, 32, 2, 0, 0, 1, 4>(p, s);            // 128x128 ring depths for the paired-split bound (with SPLIT=2, FP8MI_DEBUG=1): 2 x 32 KiB (the shipped 128x128)
    case 131: return launch<128, 128, 64, 32, 4, 1, 0, 1, 4>(p, s);            //   4 x 32 KiB
    case 132: return launch<128, 128, 64, 32, 2, 1, 0, 2, 4>(p, s);            //   2 x 64 KiB (two K-steps per stage)
    case 133: return launch<128, 128, 64, 32, 3, 1, 0, 1, 4>(p, s);            //   3 x 32 KiB
    case 134: return launch<128, 128, 64, 32, 4, 1, 1, 1, 4>(p, s);            //   4 x 32 KiB + L2 prefetch
    case 135: return launch<128, 128, 64, 32, 3, 1, 1, 1, 4>(p, s);            //   3 x 32 KiB + L2 prefetch
    case 136: return launch<128, 128, 64, 32, 4, 0, 0, 1, 4>(p, s);            //   4 x 32 KiB, stage DMA first
    case 140: return launch<128, 64, 32, 32, 3, 1, 9, 2, 4>(p, s);             // 128x64 as shipped + A-panel prefetch by one more wave
    case 141: return launch<128, 64, 32, 32, 3, 1, 10, 2, 4>(p, s);            //   ... both two stages ahead
    case 150: return launch<64, 64, 32, 32, 3, 1, 0, 2, 4>(p, s);              // 64x64 tiles, 4 waves, 3 x 32 KiB ring (decode regime: M <= 64 with a 4-way K split instead of 64x128 x 8)
    case 151: return launch<64, 64, 32, 32, 4, 1, 0, 2, 4>(p, s);              //   4 x 32 KiB
    case 152: return launch<64, 64, 16, 32, 4, 1, 0, 2, 4>(p, s);              //   8 waves
    case 153: return launch<32, 128, 16, 32, 3, 1, 0, 2, 4>(p, s);             // 32x128, 8 waves of 16x32, 3 x 40 KiB (M <= 32)
    case 154: return launch<32, 64, 16, 32, 4, 1, 0, 2, 4>(p, s);              // 32x64, 4 waves, 4 x 24 KiB
    case 155: return launch<16, 128, 16, 32, 4, 1, 0, 2, 2>(p, s);             // 16x128, 4 waves (2 loading), 4 x 36 KiB (M <= 16)
    case 137: return launch<64, 64, 16, 32, 4, 1, 64, 2, 4>(p, s);             // 64x64 as shipped, B operand through plain loads into a register sink: TIMING ONLY (wrong results)
    case 138: return launch<32, 64, 16, 32, 4, 1, 64, 2, 4>(p, s);             //   ... 32x64
    case 144: return launch<64, 64, 16, 32, 4, 1, 128, 2, 4>(p, s);            // 64x64 as shipped, a tile's K slices on one XCD and the exchange at group scope (Cfg::XLOCAL): what relying on the placement would buy
    case 139: return launch<64, 64, 16, 32, 4, 1, 16, 2, 4>(p, s);             // 64x64 as shipped, timing-only floors (Cfg::FLOOR, with the split-K exchange the launch resolves): the DMA stream only
    case 142: return launch<64, 64, 16, 32, 4, 1, 32, 2, 4>(p, s);             //   ... no K loop: launch + split-K exchange + epilogue
    case 143: return launch<64, 64, 16, 32, 4, 1, 48, 2, 4>(p, s);             //   ... the launch alone
    case 156: return launch<64, 64, 16, 32, 3, 1, 0, 2, 4>(p, s);              // 64x64, 8 waves, 3 x 32 KiB
    case 157: return launch<64, 64, 16, 32, 2, 1, 0, 2, 4>(p, s);              // 64x64, 8 waves, 2 x 32 KiB (two workgroups per CU)
    case 126: return launch<64, 16, 16, 16, 4, 1, 0, 2, 2>(p, s);              // 64x16, 4 waves (2 loading), 4 x 20 KiB: N / 16 tiles fill the chip at N = 4096 with NO K split (fp32 out only: timing experiment)
    case 127: return launch<64, 16, 16, 16, 6, 1, 0, 2, 2>(p, s);              //   6 x 20 KiB
    case 128: return launch<32, 16, 16, 16, 6, 1, 0, 2, 2>(p, s);              // 32x16, 2 waves
    case 158: return launch<64, 32, 16, 32, 4, 1, 0, 2, 4>(p, s);              // 64x32, 4 waves, 4 x 24 KiB (more tiles -> fewer K slices, smaller partials)
    case 159: return launch<32, 32, 16, 32, 4, 1, 0, 2, 2>(p, s);              // 32x32, 2 waves, 4 x 16 KiB
    case 7: return launch<128, 64, 64, 32, 6>(p, s);                           // 128x64, 4 waves
    case 8: return launch<128, 128, 64, 64, 4>(p, s);                          // 128x128, 4 waves, 4-stage ring
    case 9: return launch<256, 128, 64, 64, 3, 0, 0, 1, 4>(p, s);              // 256x128, 8 waves (0-3 load), 3 x 48 KiB
    case 10: return launch<128, 64, 32, 32, 6>(p, s);                          // 128x64, 8 waves, one K-step per stage (6 x 24 KiB)
    case 11: return launch<256, 256, 128, 64, 2>(p, s);                        // 256x256, all 8 waves load
    case 12: return launch<128, 64, 32, 32, 3, 0, 0, 2>(p, s);                 // 128x64 KS=2, all 8 waves load
    case 13: return launch<128, 128, 64, 32, 2>(p, s);                         // 128x128, all 8 waves load
#endif
    default: return FP8MI_E_ENUM;
    }
}
