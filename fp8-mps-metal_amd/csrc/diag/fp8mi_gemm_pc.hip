// Producer / consumer FP8 e4m3fn GEMM for gfx950 - the tile kernel for problems whose tile grid gives each CU ONE
// workgroup (M x N up to a few hundred tiles: BASELINE config C3 M=512 K=N=4096, the column shards of a sharded
// linear, small-batch decode with split-K).  Same math, operand layout, LDS image, swizzle and fused epilogue as the
// symmetric ring kernel (fp8mi_gemm.hip); what differs is who does what:
//
//   * In-kernel stamps of the symmetric kernel on C3 (profiles/r02_stamps.txt) showed where its K loop goes: a
//     `buffer_load ... lds` instruction holds its wave ~64 cycles, because the CU's one texture-address path takes
//     1 KiB per 16 cycles and four waves share it.  A wave that issues the stage DMA (12 instructions, ~800 cycles)
//     and THEN computes (~520 cycles) is the critical path of every stage while its SIMD partner, done after 560
//     cycles, waits at the barrier: 1,576 cycles per stage for 512 cycles of MFMA work per SIMD.
//   * Here the workgroup has two kinds of waves.  LOADER waves (one per SIMD) do nothing but issue the LDS-DMA of
//     stage t + NSTAGE - 1, wait (counted vmcnt) until stage t has landed and meet the barrier: the address path is
//     fed back to back and a blocked issue blocks nobody's MFMAs.  CONSUMER waves (one or two per SIMD) never touch
//     global memory in the loop: they read MFMA fragments from LDS into one of TWO register sets and multiply from the
//     other, so the ds_read latency of K-step u + 1 hides under the MFMAs of K-step u (the symmetric kernel cannot
//     afford the second set: all of its waves carry accumulators AND staging state).
//   * One raw s_barrier per stage (KS K-steps) for all waves; a consumer reaches it having finished its LDS reads of
//     the previous stage (lgkmcnt(0)), which is what lets the loaders overwrite that slot right after the barrier.
//
// Replaces fp8_scaled_matmul_kernel (fp8_matmul.metal:99-147) on these shapes.  NaN bytes, split-K and the epilogue
// work as in fp8mi_gemm.hip (NaN accumulator -> the tile is recomputed with scrubbed fragments; last-arriving slice
// sums the partials in slice order; (acc*sa)*sb + bias, * scale_result, cast).

// DIAGNOSTIC LIBRARY ONLY (make libfp8mi_diag.so): measured on MI355X this kernel is parity-clean but not faster than the
// ring kernel on any shape (C3 16.8 vs 15.3 us) - and its timing-only ablations say why: the LDS-DMA stream ALONE takes the
// whole 16.4 us (221 MB through the CUs' address paths at ~24 TB/s), MFMAs and fragment reads hide under it completely
// (DESIGN.md 6).  It stays in the tree as the instrument that measured that.
#ifdef FP8MI_DIAG

#include "fp8mi_gemm_epi.h"

namespace {

template <int BM_, int BN_, int CWM_, int CWN_, int NL_, int NSTAGE_, int KS_, int ABL_ = 0>
struct PCfg {
    static constexpr int ABL = ABL_;  // timing-only ablations (diagnostic build): 1 no ds_read, 2 no MFMA, 4 no LDS-DMA
    static constexpr int BM = BM_, BN = BN_;
    static constexpr int kWavesM = CWM_, kWavesN = CWN_;
    static constexpr int kConsumers = CWM_ * CWN_;
    static constexpr int kLoaders = NL_;
    static constexpr int kWaves = kConsumers;            // the waves that hold accumulators (what the epilogues index by)
    static constexpr int kAllWaves = kConsumers + NL_;
    static constexpr int kThreads = kAllWaves * 64;
    static constexpr int kCThreads = kConsumers * 64;
    static constexpr int WM = BM_ / CWM_, WN = BN_ / CWN_;
    static constexpr int TM = WM / 16, TN = WN / 16;
    static constexpr int KS = KS_, NSTAGE = NSTAGE_, PF = NSTAGE_ - 1;
    static constexpr int kGroupsA = BM_ / 8, kGroupsB = BN_ / 8, kGroups = kGroupsA + kGroupsB;  // 1-KiB staging groups per K-step
    static constexpr int kStepBytes = (BM_ + BN_) * BK;
    static constexpr int kStageBytes = KS_ * kStepBytes;
    static constexpr int kRingBytes = NSTAGE_ * kStageBytes;
    static constexpr int JA = kGroupsA / NL_, JB = kGroupsB / NL_;
    static constexpr int kDma = KS_ * (JA + JB);         // LDS-DMA instructions per stage per loader wave
    static_assert(WM % 16 == 0 && WN % 16 == 0 && TM >= 1 && TN >= 1, "whole MFMA fragments per consumer");
    static_assert(kGroupsA % NL_ == 0 && kGroupsB % NL_ == 0 && NL_ % 2 == 0, "loaders stage whole groups; even count for the shared swizzle");
    static_assert(NSTAGE_ >= 3 && NSTAGE_ <= 6 && (NSTAGE_ - 1) * kDma <= 63, "vmcnt is a 6-bit counter");
    static_assert(kRingBytes + 16 <= 160 * 1024, "LDS is 160 KiB per CU");
};

typedef __attribute__((address_space(3))) uint8_t lds_u8;  // explicit LDS pointers: ds_read / M0 bases without flat-pointer casts

template <int N>
FP8MI_DEVICE void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// this wave's LDS-DMA of all but the `newer` youngest stages has landed
template <typename C>
FP8MI_DEVICE void wait_landed(int newer)
{
    constexpr int G = C::kDma;
    if (4 * G <= 63 && newer >= 4) wait_vm<(4 * G <= 63 ? 4 * G : 0)>();
    else if (3 * G <= 63 && newer == 3) wait_vm<(3 * G <= 63 ? 3 * G : 0)>();
    else if (2 * G <= 63 && newer == 2) wait_vm<(2 * G <= 63 ? 2 * G : 0)>();
    else if (newer == 1) wait_vm<G>();
    else wait_vm<0>();
}

template <typename C>
struct LoadPlan {
    uint32_t va0, vb0;   // byte offset of this lane's 16 bytes in the wave's first A / B group (k = 0)
    uint32_t row0;       // its row there
    uint32_t kpos;       // position inside the 128-byte K-step
    uint32_t sa, sb;     // byte stride between the wave's consecutive A / B groups
    int rows_a, rows_b;  // valid rows of this tile
    bool full;
};

// one ring stage (KS K-steps of A and B rows) -> LDS.  k0 = first k of the stage.
template <typename C, bool TAIL>
FP8MI_DEVICE void issue_stage(const LoadPlan<C> &pl, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, lds_u8 *stage,
                              int lw, int k0, int64_t K)
{
#pragma unroll
    for (int q = 0; q < C::KS; ++q) {
#pragma unroll
        for (int j = 0; j < C::JA + C::JB; ++j) {
            const bool is_a = j < C::JA;
            const int jo = is_a ? j : j - C::JA;
            const int gs = q * C::kGroups + (is_a ? 0 : C::kGroupsA) + lw + jo * C::kLoaders;  // LDS group slot
            uint32_t vo = (is_a ? pl.va0 + jo * pl.sa : pl.vb0 + jo * pl.sb) + q * BK;
            if (!pl.full && (int)(pl.row0 + jo * C::kLoaders * 8) >= (is_a ? pl.rows_a : pl.rows_b)) vo = kOOB;
            if (TAIL && (int64_t)k0 + q * BK + pl.kpos >= K) vo = kOOB;   // K tail (and a padding K-step past K)
            lds_void *dst = (lds_void *)(stage + gs * 1024);
            if constexpr (C::ABL & 4) { asm volatile("" ::"v"(vo), "s"(dst)); continue; }
            if (is_a) __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, dst, 16, (int)vo, k0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, dst, 16, (int)vo, k0, 0, 0);
        }
    }
}

// LOADER wave: stages s0 .. s0 + ns - 1 of this workgroup's K range through the ring
template <typename C>
FP8MI_DEVICE void load_loop(const LoadPlan<C> &pl, __amdgpu_buffer_rsrc_t ra, __amdgpu_buffer_rsrc_t rb, lds_u8 *smem, int lw,
                            int s0, int ns, int64_t K)
{
    constexpr int kStageK = BK * C::KS;
    const int last = (int)((K + kStageK - 1) / kStageK) - 1;  // the stage that may hold the K tail
    auto issue = [&](int s, int slot) {
        const int sg = s0 + s;
        if (sg >= last) issue_stage<C, true>(pl, ra, rb, smem + slot * C::kStageBytes, lw, sg * kStageK, K);
        else issue_stage<C, false>(pl, ra, rb, smem + slot * C::kStageBytes, lw, sg * kStageK, K);
    };
#pragma unroll
    for (int s = 0; s < C::PF; ++s)
        if (s < ns) issue(s, s);
    int fill = C::PF % C::NSTAGE;
    for (int t = 0; t < ns; ++t) {
        wait_landed<C>(min(C::PF - 1, ns - 1 - t));  // stage t: this wave's part is in LDS
        __builtin_amdgcn_s_barrier();                 // #t: ... everybody's; and every consumer has read stage t - 1
        if (t + C::PF < ns) issue(t + C::PF, fill);   // into the slot of stage t - 1
        fill = (fill + 1 == C::NSTAGE) ? 0 : fill + 1;
    }
}

template <typename C>
FP8MI_DEVICE void read_frags(const lds_u8 *unit, int a_row0, int b_row0, uint32_t off1, uint32_t off2, i32x8 (&xf)[C::TM],
                             i32x8 (&wf)[C::TN])
{
    const lds_u8 *sa = unit + a_row0 * BK;                       // A rows (m)
    const lds_u8 *sb = unit + C::kGroupsA * 1024 + b_row0 * BK;  // B rows (n)
    if constexpr (C::ABL & 1) {
        const int j = (int)(uintptr_t)unit + (int)off1;
#pragma unroll
        for (int t = 0; t < C::TN; ++t) wf[t] = i32x8{j, 1, j, 1, j, 1, j, 1};
#pragma unroll
        for (int t = 0; t < C::TM; ++t) xf[t] = i32x8{j, j, j, j, j, j, j, j};
        return;
    }
#pragma unroll
    for (int t = 0; t < C::TN; ++t) {
        i32x4 lo = *(const __attribute__((address_space(3))) i32x4 *)(sb + t * 16 * BK + off1);
        i32x4 hi = *(const __attribute__((address_space(3))) i32x4 *)(sb + t * 16 * BK + off2);
        wf[t] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int t = 0; t < C::TM; ++t) {
        i32x4 lo = *(const __attribute__((address_space(3))) i32x4 *)(sa + t * 16 * BK + off1);
        i32x4 hi = *(const __attribute__((address_space(3))) i32x4 *)(sa + t * 16 * BK + off2);
        xf[t] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
}

template <typename C, bool SCRUB>
FP8MI_DEVICE void mfma_unit(i32x8 (&xf)[C::TM], i32x8 (&wf)[C::TN], f32x4 (&acc)[C::TN][C::TM])
{
    if constexpr (SCRUB) {  // reference NaN-byte semantics, redo pass only (fp8_matmul.metal:21); in place: the set is refilled next
#pragma unroll
        for (int t = 0; t < C::TM; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) xf[t][j] = (int)scrub_nan4((uint32_t)xf[t][j]);
#pragma unroll
        for (int t = 0; t < C::TN; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) wf[t][j] = (int)scrub_nan4((uint32_t)wf[t][j]);
    }
    if constexpr (C::ABL & 2) {
#pragma unroll
        for (int t = 0; t < C::TM; ++t) asm volatile("" ::"v"(xf[t]));
#pragma unroll
        for (int t = 0; t < C::TN; ++t) asm volatile("" ::"v"(wf[t]));
        return;
    }
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm)
            acc[tn][tm] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[tn], xf[tm], acc[tn][tm], 0, 0, 0, kScaleOne, 0, kScaleOne);
}

// hipcc treats an MFMA as a pure register computation and moves it across raw barriers and asm waits: left alone it
// sank the MFMAs of K-step u below the barrier and the reads of K-step u + 2 (a third register set, and the LDS
// latency back on the critical path).  A scheduling fence keeps each block where it is written.
#define FP8MI_PC_FENCE() __builtin_amdgcn_sched_barrier(0)

// ... and the IR-level passes sink the (pure) MFMA calls of K-step u past the conditional barrier block to their next
// use.  Making the accumulators opaque right after the block that produced them pins that block in program order
// (no instruction is emitted).
template <typename C>
FP8MI_DEVICE void pin_acc(f32x4 (&acc)[C::TN][C::TM])
{
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm) asm volatile("" : "+v"(acc[tn][tm]));
}

// CONSUMER wave: ns stages = ns * KS K-steps.  K-step u + 1 is fetched into one register set (A / B) while K-step u is
// multiplied from the other; the loop body is one PAIR of K-steps so that both sets have fixed names.  A stage
// boundary costs the wave one `s_waitcnt lgkmcnt(0)` (its reads of the old stage are done: the loaders may overwrite
// that slot after the barrier) and the barrier (the next stage has landed).
template <typename C, bool SCRUB>
FP8MI_DEVICE void consume_loop(const lds_u8 *smem, int wm0, int wn0, uint32_t off1, uint32_t off2, int ns,
                               f32x4 (&acc)[C::TN][C::TM])
{
    static_assert(C::KS == 1 || C::KS == 2, "the pair loop below is written for one or two K-steps per stage");
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm) acc[tn][tm] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (ns == 0) return;
    i32x8 xa[C::TM], wa[C::TN], xb[C::TM], wb[C::TN];
    const lds_u8 *cur = smem;  // slot of the stage being read
    int slot = 0;
    auto next_stage = [&]() {
        slot = (slot + 1 == C::NSTAGE) ? 0 : slot + 1;
        cur = smem + slot * C::kStageBytes;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    __builtin_amdgcn_s_barrier();  // #0: stage 0 has landed
    read_frags<C>(cur, wm0, wn0, off1, off2, xa, wa);
    if constexpr (C::KS == 2) {
        for (int s = 0; s < ns; ++s) {
            read_frags<C>(cur + C::kStepBytes, wm0, wn0, off1, off2, xb, wb);
            FP8MI_PC_FENCE();
            mfma_unit<C, SCRUB>(xa, wa, acc);
            pin_acc<C>(acc);
            FP8MI_PC_FENCE();
            if (s + 1 < ns) {
                next_stage();
                read_frags<C>(cur, wm0, wn0, off1, off2, xa, wa);
            }
            FP8MI_PC_FENCE();
            mfma_unit<C, SCRUB>(xb, wb, acc);
            pin_acc<C>(acc);
            FP8MI_PC_FENCE();
        }
    } else {
        int s = 0;
        for (; s + 1 < ns; s += 2) {
            next_stage();
            read_frags<C>(cur, wm0, wn0, off1, off2, xb, wb);
            FP8MI_PC_FENCE();
            mfma_unit<C, SCRUB>(xa, wa, acc);
            pin_acc<C>(acc);
            FP8MI_PC_FENCE();
            if (s + 2 < ns) {
                next_stage();
                read_frags<C>(cur, wm0, wn0, off1, off2, xa, wa);
            }
            FP8MI_PC_FENCE();
            mfma_unit<C, SCRUB>(xb, wb, acc);
            pin_acc<C>(acc);
            FP8MI_PC_FENCE();
        }
        if (s < ns) mfma_unit<C, SCRUB>(xa, wa, acc);  // odd stage count: the last stage is in A
    }
}

template <int BM, int BN, int CWM, int CWN, int NL, int NSTAGE, int KS, int ABL>
__global__ __launch_bounds__((PCfg<BM, BN, CWM, CWN, NL, NSTAGE, KS, ABL>::kThreads)) void gemm_pc_kernel(MMParams p_in, int tiles_m, int tiles_n,
                                                                                                    int vec_store, int nwg)
{
    using C = PCfg<BM, BN, CWM, CWN, NL, NSTAGE, KS, ABL>;
    __shared__ __attribute__((aligned(16))) uint8_t smem[C::kRingBytes + kFlagBytes];
    if (threadIdx.x == 0) *(volatile int *)(smem + C::kRingBytes) = 0;  // NaN verdict word (ordered by the K loop's barriers)
    const MMParams p = pin_params(p_in);  // every kernel argument in one scalar-load clause (fp8mi_common.h)
    FP8MI_PIN_S(tiles_m); FP8MI_PIN_S(tiles_n); FP8MI_PIN_S(vec_store); FP8MI_PIN_S(nwg);
    const EpiScalars es = load_epi_scalars(p);  // in flight under the K loop

    int tile_m, tile_n, kslice, wg;
    tile_of_block(blockIdx.x, nwg, tiles_m, tiles_n, tile_m, tile_n, kslice, wg);
    const int64_t m0 = (int64_t)tile_m * BM, n0 = (int64_t)tile_n * BN;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool is_loader = wave >= C::kConsumers;  // wave-uniform
    const int lw = wave - C::kConsumers;

    const int64_t rows_a = min((int64_t)BM, p.M - m0), rows_b = min((int64_t)BN, p.N - n0);
    constexpr int kStageK = BK * C::KS;
    const int ns_all = (int)((p.K + kStageK - 1) / kStageK);
    const int nsplit = p.split > 1 ? p.split : 1;
    const int ns_slice = (ns_all + nsplit - 1) / nsplit;  // the host made every slice non-empty
    const int s0 = kslice * ns_slice, ns = min(ns_slice, ns_all - s0);
    // the ring as an LDS-address-space pointer (ds_read / M0 bases without flat-pointer casts)
    lds_u8 *ring = (lds_u8 *)smem;

    // Reference NaN-byte semantics (fp8_matmul.metal:21) without paying for them: a NaN accumulator proves a NaN byte
    // took part (fp8mi_gemm_epi.h); the consumers raise the verdict word before the barrier that ends the K loop, every
    // wave reads it behind that barrier, and only then the tile is redone with scrubbed fragments (pass 1).
    volatile int *flag = (volatile int *)(smem + C::kRingBytes);

    if (is_loader) {
        // ---- LOADER waves: everything they need lives only on this path (the register allocation is per kernel:
        //      loop-invariant staging addresses computed ahead of the branch would be carried by the consumers too)
        const int64_t bytes_a = (rows_a - 1) * p.lda + p.K, bytes_b = (rows_b - 1) * p.ldb + p.K;
        __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)(p.A + m0 * p.lda), 0,
                                                                       (int)min(bytes_a, (int64_t)0x7FFFFFFF), 0x00020000);
        __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void *)(p.B + n0 * p.ldb), 0,
                                                                       (int)min(bytes_b, (int64_t)0x7FFFFFFF), 0x00020000);
        LoadPlan<C> pl;
        const int row0 = lw * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ (((lw & 1) * 4 + (lane >> 4)) & 7);  // = (lane & 7) ^ ((row >> 1) & 7) for each of the wave's groups
        pl.row0 = (uint32_t)row0;
        pl.kpos = (uint32_t)(chunk * 16);
        pl.va0 = (uint32_t)(row0 * p.lda + chunk * 16);
        pl.vb0 = (uint32_t)(row0 * p.ldb + chunk * 16);
        pl.sa = (uint32_t)(C::kLoaders * 8 * p.lda);
        pl.sb = (uint32_t)(C::kLoaders * 8 * p.ldb);
        pl.rows_a = (int)rows_a;
        pl.rows_b = (int)rows_b;
        pl.full = rows_a == BM && rows_b == BN;
        for (int pass = 0; pass < 2; ++pass) {
            load_loop<C>(pl, ra, rb, ring, lw, s0, ns, p.K);
            __builtin_amdgcn_s_barrier();  // the ring is free; the verdict is in
            if (!p.nan_zero || pass == 1 || !*flag) break;
        }
        return;  // (s_barrier counts only the waves that are still alive)
    }

    // ---- CONSUMER waves
    const int wm0 = (wave % C::kWavesM) * C::WM, wn0 = (wave / C::kWavesM) * C::WN;
    const int fr = lane & 15, fg = lane >> 4;
    const uint32_t off1 = (uint32_t)(fr * BK + ((fg ^ (fr >> 1)) << 4));
    const uint32_t off2 = (uint32_t)(fr * BK + (((4 + fg) ^ (fr >> 1)) << 4));
    f32x4 acc[C::TN][C::TM];
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 0) consume_loop<C, false>(ring, wm0, wn0, off1, off2, ns, acc);
        else consume_loop<C, true>(ring, wm0, wn0, off1, off2, ns, acc);
        if (p.nan_zero && pass == 0 && acc_has_nan<C>(acc)) *flag = 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // the ring is free (staging buffer of the epilogue); the verdict is in
        if (!p.nan_zero || pass == 1 || !*flag) break;
    }

    if (nsplit > 1) {
        if (!splitk_combine<C>(p, acc, smem, wg, kslice, nsplit, tiles_m * tiles_n)) return;
    }

    const int rows_m = (int)rows_a, cols_n = (int)rows_b;
    if (rows_m == BM && cols_n == BN && vec_store) {
        if (p.out_dtype == FP8MI_F32) epilogue_staged<C, FP8MI_F32>(p, es, acc, smem, m0, n0, wave, wm0, wn0, lane);
        else if (p.out_dtype == FP8MI_BF16) epilogue_staged<C, FP8MI_BF16>(p, es, acc, smem, m0, n0, wave, wm0, wn0, lane);
        else epilogue_staged<C, FP8MI_F16>(p, es, acc, smem, m0, n0, wave, wm0, wn0, lane);
    } else if (p.out_dtype == FP8MI_F32) epilogue<C, FP8MI_F32>(p, es, acc, m0, n0, wm0, wn0, fr, fg, rows_m, cols_n, vec_store);
    else if (p.out_dtype == FP8MI_BF16) epilogue<C, FP8MI_BF16>(p, es, acc, m0, n0, wm0, wn0, fr, fg, rows_m, cols_n, vec_store);
    else epilogue<C, FP8MI_F16>(p, es, acc, m0, n0, wm0, wn0, fr, fg, rows_m, cols_n, vec_store);
}

template <int BM, int BN, int CWM, int CWN, int NL, int NSTAGE, int KS, int ABL = 0>
int launch_pc(const MMParams &p_in, hipStream_t s)
{
    using C = PCfg<BM, BN, CWM, CWN, NL, NSTAGE, KS, ABL>;
    MMParams p = p_in;
    const int64_t tm = (p.M + BM - 1) / BM, tn = (p.N + BN - 1) / BN;
    if (tm * tn > 0x7FFFFFFF) return FP8MI_E_UNSUPPORTED;
    const int rc = resolve_split(p, tm, tn, BM, BN, BK * KS);
    if (rc) return rc;
    const int esz = p.out_dtype == FP8MI_F32 ? 4 : 2;
    const int vec = (((p.ldc * esz) % 16) == 0 && (((uintptr_t)p.C) % 16) == 0) ? 1 : 0;
    const unsigned grid = (unsigned)(tm * tn * p.split);
    return fp8mi_launch(gemm_pc_kernel<BM, BN, CWM, CWN, NL, NSTAGE, KS, ABL>, dim3(grid), dim3(C::kThreads), s, p, (int)tm, (int)tn, vec, (int)grid);
}

}  // namespace

int fp8mi_launch_gemm_pc(const MMParams &p, int variant, hipStream_t s)
{
    switch (variant) {
    case 15: return launch_pc<128, 64, 2, 2, 4, 3, 2>(p, s);    // 4 consumers (64x32) + 4 loaders, 3 x 48 KiB
    case 16: return launch_pc<64, 128, 2, 2, 4, 3, 2>(p, s);    // 4 consumers (32x64) + 4 loaders, for M <= 64
    case 17: return launch_pc<128, 128, 2, 2, 4, 4, 1>(p, s);  // 4 consumers (64x64) + 4 loaders, 4 x 32 KiB
    // A/B variants (diagnostic ids)
    case 20: return launch_pc<128, 64, 4, 2, 4, 3, 2>(p, s);     // 8 consumers (32x32) + 4 loaders
    case 21: return launch_pc<128, 128, 4, 2, 4, 4, 1>(p, s);    // 8 consumers (32x64) + 4 loaders
    case 22: return launch_pc<128, 64, 2, 2, 2, 3, 2>(p, s);     // 4 consumers + 2 loaders
    case 24: return launch_pc<128, 64, 2, 2, 4, 6, 1>(p, s);     // one K-step per stage, 6 x 24 KiB
    case 25: return launch_pc<128, 64, 2, 2, 8, 3, 2>(p, s);     // 4 consumers + 8 loaders (two per SIMD)
    case 26: return launch_pc<128, 64, 4, 2, 8, 3, 2>(p, s);     // 8 consumers + 8 loaders
    case 27: return launch_pc<128, 128, 2, 2, 8, 4, 1>(p, s);    // 128x128, 4 consumers + 8 loaders
    case 28: return launch_pc<128, 64, 2, 2, 8, 6, 1>(p, s);     // 8 loaders, one K-step per stage
    case 208: return launch_pc<128, 64, 2, 2, 8, 3, 2, 3>(p, s); // LDS-DMA only, 8 loaders (timing only)
    // timing-only ablations of the 128x64 kernel (results are garbage): 20x, x = ablation bits
    case 201: return launch_pc<128, 64, 2, 2, 4, 3, 2, 1>(p, s);   // no ds_read
    case 202: return launch_pc<128, 64, 2, 2, 4, 3, 2, 2>(p, s);   // no MFMA
    case 203: return launch_pc<128, 64, 2, 2, 4, 3, 2, 3>(p, s);   // LDS-DMA only
    case 204: return launch_pc<128, 64, 2, 2, 4, 3, 2, 4>(p, s);   // no LDS-DMA
    case 205: return launch_pc<128, 64, 2, 2, 4, 3, 2, 5>(p, s);   // MFMA only
    case 206: return launch_pc<128, 64, 2, 2, 4, 3, 2, 6>(p, s);   // ds_read only
    case 207: return launch_pc<128, 64, 2, 2, 4, 3, 2, 7>(p, s);   // barriers only
    default: return FP8MI_E_ENUM;
    }
}

#endif  // FP8MI_DIAG
