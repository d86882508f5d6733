// Device code shared by the tiled FP8 GEMM kernels (fp8mi_gemm.hip: symmetric ring kernel;
// fp8mi_gemm_pc.hip: producer / consumer kernel): constants of the 128-byte K-step and the fused epilogues.
#pragma once

#include "fp8mi_common.h"

namespace {

constexpr int BK = 128;  // bytes (= k elements) per K-step
constexpr uint32_t kOOB = 0x80000000u;
constexpr int kScaleOne = 0x7F7F7F7F;  // E8M0 127 = 2^0 in every byte
static_assert(kWsCounterBytes == FP8MI_WS_COUNTER_BYTES, "include/fp8mi.h and the kernels agree on the counter block");

typedef __attribute__((address_space(3))) void lds_void;

// The per-tensor epilogue scalars are fetched at kernel ENTRY (scalar loads that complete under the K loop): loaded where
// they are used, each was a dependent global load - ~1 us of latency between the last MFMA and the first store.
struct EpiScalars {
    float sa0, sb0, sr;
};

FP8MI_DEVICE EpiScalars load_epi_scalars(const MMParams &p)
{
    EpiScalars e;
    e.sa0 = p.scale_a[0];
    e.sb0 = p.scale_b[0];
    e.sr = p.scale_result ? p.scale_result[0] : 1.0f;
    return e;
}

// Fused epilogue, in the reference's order (fp8_matmul.metal:144-146, then
// fp8_mps_patch.py:94-104): (acc * sa) * sb, + bias, * scale_result, cast.
// Lane (fr = lane & 15, fg = lane >> 4) holds, per 16x16 fragment (tn, tm), the
// 4 consecutive columns n = tn*16 + 4 fg + j of row m = tm*16 + fr: one 16-byte
// (fp32) or 8-byte (bf16 / f16) store.  Column scales and bias are loaded once
// per lane, row scales once per fragment row; everything else is 32-bit math.
template <typename C, int OUT>
FP8MI_DEVICE void epilogue(const MMParams &p, const EpiScalars &es, const f32x4 (&acc)[C::TN][C::TM], int64_t m0, int64_t n0, int wm0,
                           int wn0, int fr, int fg, int rows_m, int cols_n, int vec_store)
{
    const bool has_bias = p.bias != nullptr;
    const float sr = es.sr;
    const bool has_sr = p.scale_result != nullptr;
    float sbv[C::TN][4], bv[C::TN][4];
    const float sb0 = es.sb0;
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int nl = min(wn0 + tn * 16 + fg * 4 + j, cols_n - 1);  // clamped: the value is unused past the edge
            sbv[tn][j] = p.sb_row ? p.scale_b[n0 + nl] : sb0;
            bv[tn][j] = (has_bias && !p.transposed) ? load_as_float(p.bias, n0 + nl, p.bias_dtype) : 0.0f;
        }
    const float sa0 = es.sa0;
    constexpr int kEsz = OUT == FP8MI_F32 ? 4 : 2;
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm) {
        const int ml = wm0 + tm * 16 + fr;
        if (ml >= rows_m) continue;
        const float sa = p.sa_row ? p.scale_a[m0 + ml] : sa0;
        const float brow = (has_bias && p.transposed) ? load_as_float(p.bias, m0 + ml, p.bias_dtype) : 0.0f;
        uint8_t *row = (uint8_t *)p.C + ((m0 + ml) * p.ldc + n0) * kEsz;
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn) {
            const int nl = wn0 + tn * 16 + fg * 4;
            if (nl >= cols_n) continue;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float r = p.transposed ? (acc[tn][tm][j] * sbv[tn][j]) * sa : (acc[tn][tm][j] * sa) * sbv[tn][j];
                if (has_bias) r = r + (p.transposed ? brow : bv[tn][j]);
                if (has_sr) r = r * sr;
                v[j] = r;
            }
            uint8_t *dst = row + nl * kEsz;
            if (vec_store && nl + 3 < cols_n) {
                if (OUT == FP8MI_F32) {
                    *(f32x4 *)dst = f32x4{v[0], v[1], v[2], v[3]};
                } else if (OUT == FP8MI_BF16) {
                    __bf16 h0 = (__bf16)v[0], h1 = (__bf16)v[1], h2 = (__bf16)v[2], h3 = (__bf16)v[3];
                    *(u32x2 *)dst = u32x2{(uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16),
                                          (uint32_t)__builtin_bit_cast(uint16_t, h2) | ((uint32_t)__builtin_bit_cast(uint16_t, h3) << 16)};
                } else {
                    _Float16 h0 = (_Float16)v[0], h1 = (_Float16)v[1], h2 = (_Float16)v[2], h3 = (_Float16)v[3];
                    *(u32x2 *)dst = u32x2{(uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16),
                                          (uint32_t)__builtin_bit_cast(uint16_t, h2) | ((uint32_t)__builtin_bit_cast(uint16_t, h3) << 16)};
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (nl + j < cols_n) {
                        if (OUT == FP8MI_F32) ((float *)dst)[j] = v[j];
                        else if (OUT == FP8MI_BF16) ((__bf16 *)dst)[j] = (__bf16)v[j];
                        else ((_Float16 *)dst)[j] = (_Float16)v[j];
                    }
                }
            }
        }
    }
}

// Epilogue for FULL tiles: same arithmetic, but each wave transposes its output
// through a private corner of the (now idle) LDS ring so that every global store
// instruction writes whole 128-byte lines (8 rows x 128 B for 16-bit outputs,
// 4 rows x 256 B for fp32) instead of 16 scattered 32- / 64-byte pieces - the
// direct form was store-issue bound (22k of 112k cycles per 256x256 bf16 tile).
template <typename C, int OUT>
FP8MI_DEVICE void epilogue_staged(const MMParams &p, const EpiScalars &es, const f32x4 (&acc)[C::TN][C::TM], uint8_t *smem, int64_t m0,
                                  int64_t n0, int wave, int wm0, int wn0, int lane)
{
    constexpr int kEsz = OUT == FP8MI_F32 ? 4 : 2;
    constexpr int WNc = C::TN * 16;              // columns of the wave tile
    constexpr int kRowBytes = WNc * kEsz;        // 64 .. 256
    constexpr int kStride = kRowBytes + 16;      // padded: spreads the 16 rows over the banks
    constexpr int kCPR = kRowBytes / 16;         // 16-byte chunks per row
    constexpr int kRPI = 64 / kCPR;              // rows written per store instruction
    constexpr int kNI = 16 / kRPI;               // store instructions per 16-row fragment
    static_assert(C::kWaves * 16 * kStride <= C::kRingBytes, "staging fits in the ring");
    uint8_t *buf = smem + wave * (16 * kStride);
    const int fr = lane & 15, fg = lane >> 4;

    const bool has_bias = p.bias != nullptr, has_sr = p.scale_result != nullptr;
    const float sr = es.sr;
    float sbv[C::TN][4], bv[C::TN][4];
    const float sb0 = es.sb0, sa0 = es.sa0;
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = n0 + wn0 + tn * 16 + fg * 4 + j;
            sbv[tn][j] = p.sb_row ? p.scale_b[n] : sb0;
            bv[tn][j] = (has_bias && !p.transposed) ? load_as_float(p.bias, n, p.bias_dtype) : 0.0f;
        }
    const int rrow = lane / kCPR, rchunk = lane % kCPR;  // this lane's (row, 16-byte chunk) when reading back
#pragma unroll
    for (int tm = 0; tm < C::TM; ++tm) {
        const float sa = p.sa_row ? p.scale_a[m0 + wm0 + tm * 16 + fr] : sa0;
        const float brow = (has_bias && p.transposed) ? load_as_float(p.bias, m0 + wm0 + tm * 16 + fr, p.bias_dtype) : 0.0f;
#pragma unroll
        for (int tn = 0; tn < C::TN; ++tn) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float r = p.transposed ? (acc[tn][tm][j] * sbv[tn][j]) * sa : (acc[tn][tm][j] * sa) * sbv[tn][j];
                if (has_bias) r = r + (p.transposed ? brow : bv[tn][j]);
                if (has_sr) r = r * sr;
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
        // same wave wrote and reads: DS operations of one wave execute in order
        uint8_t *grow = (uint8_t *)p.C + ((m0 + wm0 + tm * 16) * p.ldc + n0 + wn0) * kEsz;
#pragma unroll
        for (int i = 0; i < kNI; ++i) {
            const int r = i * kRPI + rrow;
            u32x4 q = *(const u32x4 *)(buf + r * kStride + rchunk * 16);
            // streaming store: C is written once and not re-read by this kernel, so it should not displace the
            // A / B panels in L2 (measured: C3 -3 %, 128x128 shard -5 %, FLUX -1 %)
            store_c16(q, grow + (int64_t)r * p.ldc * kEsz + rchunk * 16);
        }
    }
}

// ---- "did a NaN byte take part in this workgroup's tile?" ---------------------------------------------------------
// Finite e4m3 products cannot overflow fp32 (|a b| <= 2e5, K <= 2^31), so a NaN accumulator proves a NaN byte took part
// (the reference decodes those to 0.0, fp8_matmul.metal:21; the matrix core propagates them).  NaN survives addition,
// and a sum of the tile's finite accumulators cannot overflow either: one packed-add tree and ONE compare per lane
// instead of a compare per element (the 512 compares + 3 barriers of the first version cost 2,400 cycles per
// 256x256 tile, in-kernel stamps).  The verdict word lives in 16 bytes of LDS behind the ring (kFlagBytes), is zeroed
// at kernel entry and read after the K loop's last barrier, which every wave passes anyway.
constexpr int kFlagBytes = 16;

template <typename C>
FP8MI_DEVICE bool acc_has_nan(const f32x4 (&acc)[C::TN][C::TM])
{
    f32x4 t = acc[0][0];
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm)
            if (tn + tm) t += acc[tn][tm];
    const float s = (t[0] + t[1]) + (t[2] + t[3]);
    return s != s;
}

// ---- XCD-aware, bijective block -> (tile, K slice) map --------------------------------------------------------
// Blocks are dealt round-robin over the 8 XCDs (b and b + 8 share one; speed only, never correctness), each XCD has
// its own L2: give every XCD a contiguous run of the work list.  The list is `split` copies of the tile grid, K slice
// slowest (the workgroups an XCD runs at one time then share one K range of A and B); inside a copy tiles go in
// groups of 4 m-tiles x all n-tiles, m fastest, so that the 32 tiles an XCD runs at one time form a 4 x 8 block whose
// A and B panels share its 4 MiB L2 instead of 16 x 2 (M=N=K=8192 bf16 492 -> 454 us; FLUX traffic 2.9x -> 2.3x).
FP8MI_DEVICE void tile_of_block(int bid, int nwg, int tiles_m, int tiles_n, int &tile_m, int &tile_n, int &kslice, int &wg)
{
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int wg_all = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int n_tiles = tiles_m * tiles_n;
    // (integer divisions cost ~150 cycles each at kernel entry, ahead of the first DMA: the common cases avoid them)
    kslice = wg_all < n_tiles ? 0 : wg_all / n_tiles;
    wg = wg_all - kslice * n_tiles;
    constexpr int kGroupM = 4;
    const int per_group = kGroupM * tiles_n;
    const int group = wg < per_group ? 0 : wg / per_group, first_m = group * kGroupM;
    const int gm = min(kGroupM, tiles_m - first_m);  // last group may be narrower: the map stays bijective
    const int in_group = wg - group * per_group;
    if (gm == kGroupM) {
        tile_m = first_m + (in_group & (kGroupM - 1));
        tile_n = in_group >> 2;
    } else {
        tile_m = first_m + in_group % gm;
        tile_n = in_group / gm;
    }
}

// ---- split-K: exchange fp32 partial tiles through the caller's workspace ----------------------------------------
// Every workgroup writes its partial tile in register order (one coalesced 16-byte store per accumulator quad), bumps
// the tile's arrival counter and leaves; the LAST arriver re-reads all slices in slice order (its own included: the
// sum does not depend on who arrived last - bit-reproducible run to run), and returns true: it runs the epilogue.
// Nobody waits.  Partials cross XCDs (private L2s), so they are written and read with sc0 sc1 (write-through /
// miss-always) accesses - publishing them with __threadfence() (whole-L2 write-back + invalidate per wave) made the
// same kernel 3-5x slower.  C::kCThreads threads (the waves that hold accumulators) call this, all of them.
template <typename C, typename = void> struct xlocal_of { static constexpr bool value = false; };   // (configurations without the knob: the diagnostic producer / consumer kernel)
template <typename C> struct xlocal_of<C, std::void_t<decltype(C::XLOCAL)>> { static constexpr bool value = C::XLOCAL; };

template <typename C>
FP8MI_DEVICE bool splitk_combine(const MMParams &p, f32x4 (&acc)[C::TN][C::TM], uint8_t *smem, int wg, int kslice, int nsplit,
                                 int n_tiles)
{
    constexpr bool kXLocal = xlocal_of<C>::value;
    constexpr int kCoherent = kXLocal ? 1 : 17;  // aux bits: sc0 | sc1 (C::XLOCAL, diagnostic timing experiment: sc0 - the XCD's L2 is the meeting point)
    int *counters = (int *)p.ws;
    constexpr int kVecPerWg = C::TN * C::TM * C::kCThreads;  // f32x4 per partial tile (register order)
    __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.ws + kWsCounterBytes), 0, (int)min(p.ws_bytes - kWsCounterBytes, (int64_t)0x7FFFFFFF), 0x00020000);
    const uint32_t slice_bytes = (uint32_t)n_tiles * kVecPerWg * 16u;
    const uint32_t my_off = ((uint32_t)wg * kVecPerWg + threadIdx.x) * 16u;  // the launcher keeps all offsets < 2^31
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[tn][tm]), rw,
                                                   (int)(my_off + (tn * C::TM + tm) * C::kCThreads * 16u),
                                                   (int)((uint32_t)kslice * slice_bytes), kCoherent);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's partial has reached memory ...
    volatile int *flag = (volatile int *)smem;
    __syncthreads();  // ... and so has every other thread's, before the workgroup's arrival is counted
    if (threadIdx.x == 0) *flag = kXLocal ? __hip_atomic_fetch_add(&counters[wg], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                                            : __hip_atomic_fetch_add(&counters[wg], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    const int arrived = *flag;
    if (arrived != nsplit - 1) return false;  // workgroup-uniform; nobody waits for anybody
    // kBatch slices' loads are in flight together (each is a miss-always read of another XCD's write: one round trip
    // per slice when issued one after the other - 8 slices cost ~5 us); the additions stay in slice order
    constexpr int kQuads = C::TN * C::TM;
    constexpr int kBatch = kQuads <= 4 ? 4 : (kQuads <= 8 ? 3 : (kQuads <= 16 ? 2 : 1));
    for (int s0 = 0; s0 < nsplit; s0 += kBatch) {
        f32x4 v[kBatch][C::TN][C::TM];
#pragma unroll
        for (int b = 0; b < kBatch; ++b) {
            const int s2 = min(s0 + b, nsplit - 1);  // past the end: re-read the last slice (not added)
#pragma unroll
            for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < C::TM; ++tm)
                    v[b][tn][tm] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        rw, (int)(my_off + (tn * C::TM + tm) * C::kCThreads * 16u), (int)((uint32_t)s2 * slice_bytes), kCoherent));
        }
#pragma unroll
        for (int b = 0; b < kBatch; ++b) {
            if (s0 + b < nsplit) {
#pragma unroll
                for (int tn = 0; tn < C::TN; ++tn)
#pragma unroll
                    for (int tm = 0; tm < C::TM; ++tm) acc[tn][tm] = (s0 + b == 0) ? v[b][tn][tm] : acc[tn][tm] + v[b][tn][tm];
            }
        }
    }
    if (threadIdx.x == 0) {  // zero for the next launch
        if (kXLocal) __hip_atomic_store(&counters[wg], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_store(&counters[wg], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();  // the flag word is part of the ring the staged epilogue reuses
    return true;
}

// ---- host: how many K slices (split-K) this launch uses ----------------------------------------------------------
// p.split on entry: 0 = automatic, 1 = none, > 1 = forced.  Automatic: when the tile grid leaves at least half of the
// device's CUs idle and K is deep, slice K so that the grid fills the chip, at least 4 ring stages per slice, at most
// 16 slices (measured: M=128 K=14336 N=4096 49 -> ~20 us).  Any request is clamped to what K and the workspace allow
// (every slice gets at least one ring stage; partials are addressed with 32-bit offsets).
inline int resolve_split(MMParams &p, int64_t tm, int64_t tn, int BM, int BN, int stage_k)
{
    const int64_t ns_all = (p.K + stage_k - 1) / stage_k;
    const int64_t cus = fp8mi_cu_count();
    int64_t split = p.split > 1 ? p.split : 1;
    if (p.split == 0 && tm * tn <= cus / 2 && ns_all >= 8) {
        split = cus / (tm * tn);
        if (split > ns_all / 4) split = ns_all / 4;
        if (split > 16) split = 16;
    }
    if (!p.ws || tm * tn > kWsCounterBytes / 4) split = 1;
    if (split > ns_all) split = ns_all > 0 ? ns_all : 1;
    if (split > 1) {
        const int64_t per_slice = tm * tn * (int64_t)BM * BN * 4;
        int64_t fit = (p.ws_bytes - kWsCounterBytes) / per_slice;
        if ((int64_t)0x7FFFFFFF / per_slice < fit) fit = (int64_t)0x7FFFFFFF / per_slice;
        if (split > fit) split = fit > 1 ? fit : 1;  // clamp to the largest count that fits
    }
    if (split > 1) {
        const int64_t per = (ns_all + split - 1) / split;
        split = (ns_all + per - 1) / per;  // drop empty slices (the kernel re-derives `per` from this count)
        if (tm * tn * split > 0x7FFFFFFF) return FP8MI_E_UNSUPPORTED;
    }
    p.split = (int)split;
    return 0;
}

}  // namespace
