// The automatic dispatch of fp8mi_scaled_mm as ONE cost model (round 4; host-only arithmetic, exported as fp8mi_choose_kernel /
// fp8mi_predict_kernel_us and tested on the CPU against measured times: tests/test_abi_and_host.py, tests/golden/dispatch_times_cold_*.json).
//
// Every kernel that takes the problem gets a predicted time in microseconds; the cheapest runs.  The forms are physical, the constants are
// FITTED (tools/dispatch_fit/, on the output of tools/sweep_regret.py: 3,871 shapes x every product kernel on MI355X, cold weights) and live in
// fp8mi_dispatch_constants.inc:
//   tile kernels   t = fixed + max(rounds x K-steps x step, operand bytes / best streaming rate) + C bytes + split-K exchange
//                  step   = max(matrix pipe, global -> LDS stream) + sync, both shared by the workgroups co-resident on a CU
//                  stream = rows staged per step x 128 B / rate(h):  1 / rate = (1 - h) / miss + h / hit, h = the share of the step's lines
//                           that another tile of the same XCD reads too (tile_of_block's 4 m-tiles x n-tiles groups); the miss rate grows
//                           as fewer CUs stream (busy ^ -e).  This is the mixed-stream ceiling of tools/probes/mlp_probe.hip
//                           (profiles/r04_mlp_probe.txt): the K loops of these kernels run AT it, which is why it predicts them.
//                  rounds = whole rounds of (CUs x workgroups per CU) + a partial round at a + (1 - a) x fill
//                  split  = resolve_split()'s own rule (fp8mi_gemm_epi.h), so the model prices the grid the launcher will really run
//   few-rows, skinny  t = fixed + max(W bytes x stream cost, K x chain cost x rounds) + per-workgroup cost
// M = 1 stays a RULE: the vec-mat, whatever a tile kernel would take (config C1-class calls keep IEEE fp32 sums for K <= 4096).
// Everything scales with the device's CU count (a 32-CU CPX partition: slots per round, streaming CUs, the whole-chip streaming rate), so the
// choice is DEFINED there, though it was fitted on 256 CUs only.
#pragma once

#include "fp8mi_common.h"

namespace fp8mi_dispatch {

struct TileCost {
    int id, bm, bn, ks, per_cu;
    bool splits;
    int64_t min_m, max_m, max_t128;
    double p[9];
};
#include "fp8mi_dispatch_constants.inc"

inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
constexpr double kDispatchFloorUs = 4.0;

// K slices the launcher will use (resolve_split in fp8mi_gemm_epi.h, restated on plain numbers; tested against it on the GPU through parity)
inline int64_t slices(const MMParams &p, const TileCost &t, int64_t tiles, int64_t cus)
{
    const int64_t ns = cdiv(p.K, 128 * t.ks);
    int64_t s = p.split > 1 ? p.split : 1;
    if (p.split == 0 && tiles <= cus / 2 && ns >= 8) {
        s = cus / tiles;
        if (s > ns / 4) s = ns / 4;
        if (s > 16) s = 16;
    }
    if (!t.splits || !p.ws || tiles > kWsCounterBytes / 4) s = 1;
    if (s > ns) s = ns > 0 ? ns : 1;
    if (s > 1) {
        const int64_t per_slice = tiles * (int64_t)t.bm * t.bn * 4;
        int64_t fit = (p.ws_bytes - kWsCounterBytes) / per_slice;
        if ((int64_t)0x7FFFFFFF / per_slice < fit) fit = (int64_t)0x7FFFFFFF / per_slice;
        if (s > fit) s = fit > 1 ? fit : 1;
    }
    if (s > 1) {
        const int64_t per = cdiv(ns, s);
        s = cdiv(ns, per);
    }
    return s < 1 ? 1 : s;
}

inline double tile_us(const MMParams &p, const TileCost &t, double cus)
{
    const double M = (double)p.M, N = (double)p.N, K = (double)p.K, esz = p.out_dtype == FP8MI_F32 ? 4.0 : 2.0;
    const int64_t tm = cdiv(p.M, t.bm), tn = cdiv(p.N, t.bn), tiles = tm * tn;
    const int64_t S = slices(p, t, tiles, (int64_t)cus), ns = cdiv(p.K, 128 * t.ks);
    const double wg = (double)tiles * (double)S, steps = (double)(cdiv(ns, S) * t.ks), slots = cus * t.per_cu;
    const double whole = (double)(int64_t)(wg / slots), fill = wg / slots - whole;
    const double vm = M / ((double)tm * t.bm), vn = N / ((double)tn * t.bn);
    const double rows = t.bm * vm + t.bn * vn;
    double conc = (wg < slots ? wg : slots) / 8.0 / (double)S;   // tiles that share one K range on an XCD at one time
    if (conc < 1.0) conc = 1.0;
    double gm = 4.0 < (double)tm ? 4.0 : (double)tm;
    if (gm > conc) gm = conc;
    double gn = conc / gm < (double)tn ? conc / gm : (double)tn;
    if (gn < 1.0) gn = 1.0;
    const double uniq = gm * t.bm * vm + gn * t.bn * vn;
    double h = 1.0 - uniq / (gm * gn * rows);
    if (h < 0.0) h = 0.0;
    double busy = wg / cus < 1.0 ? wg / cus : 1.0;
    double percu = wg / cus < 1.0 ? 1.0 : wg / cus;
    if (percu > t.per_cu) percu = t.per_cu;
    const double miss = kStreamMiss / __builtin_pow(busy > 0.125 ? busy : 0.125, kBusyExp);
    const double rate = 1.0 / ((1.0 - h) / miss + h / kStreamHit);
    const double dma = rows * 128.0 * percu / rate, pipe = t.p[1] * percu;
    const double step = (pipe > dma ? pipe : dma) + t.p[6];
    double rounds = whole + (fill > 0.0 ? t.p[2] + (1.0 - t.p[2]) * fill : 0.0);
    if (t.per_cu > 1 && whole == 0.0) rounds = rounds * t.per_cu / percu;   // a partial FIRST round: the CUs hold fewer workgroups than they could
    const double stream_floor = (M * K + N * K) / (t.p[7] * 1e6 * cus / 256.0);
    double loop = rounds * steps * step;
    if (loop < stream_floor) loop = stream_floor;
    double us = t.p[0] + loop + t.p[3] * (M * N * esz / 1e6) / (busy > 0.25 ? busy : 0.25) * (256.0 / cus);
    if (S > 1) us += t.p[4] + t.p[5] * (double)S * (t.bm * t.bn * 4.0 / 1024.0) / 64.0;
    return us > t.p[8] ? us : t.p[8];   // no launch of the kernel takes less than its own setup + one tile's epilogue
}

inline double streamer_us(const double (&c)[7], double blocks_of_x, double wg, double N, double K, double k_chain, double cus)
{
    const double mb = N * K / 1e6, stream = mb * (c[1] + c[2] * blocks_of_x) * (256.0 / cus);
    const double r = wg / (cus * c[4]), chain = (k_chain / 1e3) * (c[3] + c[5] * blocks_of_x) * (r > 1.0 ? r : 1.0);
    return c[0] + (stream > chain ? stream : chain) + c[6] * (wg / cus) * blocks_of_x;
}

inline double mx_us(const MMParams &p, double cus)
{
    const double mxp = p.M <= 2 ? 2.0 : (p.M <= 4 ? 4.0 : 8.0);
    const double rows = (mxp == 8.0 && p.K > 4096) ? 16.0 : 8.0;   // rows of W per workgroup (fp8mi_launch_gemv_mx)
    const double kp = 4096.0 * (p.K <= 4096 ? 1 : (p.K <= 8192 ? 2 : 4));   // its launch shapes hold 1, 2 or 4 wave-steps of K per wave: a K just above a step pays for the next
    return streamer_us(kMxCost, mxp / 8.0, (double)cdiv(p.N, (int64_t)rows), (double)p.N, (double)p.K, kp, cus);
}

inline double skinny_us(const MMParams &p, double cus)
{
    return streamer_us(kSkinnyCost, (double)cdiv(p.M, 16), (double)cdiv(p.N, 16), (double)p.N, (double)p.K, (double)p.K, cus);
}

// Predicted time of one kernel on this problem, or a negative value when the kernel does not take it / is not offered for it
inline double predict_us(const MMParams &p, int kernel, double cus)
{
    if (p.K <= 0) return -1.0;
    if (kernel == FP8MI_KERNEL_GEMV_MX) return fp8mi_gemv_mx_supported(p) ? mx_us(p, cus) : -1.0;
    if (kernel == FP8MI_KERNEL_SKINNY) return (p.M >= 2 && fp8mi_skinny_supported(p)) ? skinny_us(p, cus) : -1.0;
    if (!fp8mi_gemm_supported(p)) return -1.0;
    const bool w256 = fp8mi_gemm256_supported(p);
    for (const TileCost &t : kTileCosts) {
        // the 8-wave 256x256 ring kernel stands in for the one-wave-per-SIMD form where that form's envelope ends (N not a multiple of the
        // 16-byte store, K < 256): measured 15-17 % behind it (FLUX 136-141 against 118 us)
        const bool ring256 = kernel == FP8MI_KERNEL_GEMM_256 && t.id == FP8MI_KERNEL_GEMM_256W;
        if (t.id != kernel && !ring256) continue;
        if (p.M < t.min_m || p.M > t.max_m) return -1.0;
        if (t.max_t128 && cdiv(p.M, 128) * cdiv(p.N, 128) > t.max_t128) return -1.0;
        if ((t.id == FP8MI_KERNEL_GEMM_256W || t.id == FP8MI_KERNEL_GEMM_256x128W) && !ring256 && !w256) return -1.0;
        if (ring256 && w256) return -1.0;
        return tile_us(p, t, cus) * (ring256 ? 1.17 : 1.0);
    }
    return -1.0;
}

inline int choose(const MMParams &p, double cus)
{
    if (fp8mi_gemv_supported(p)) return FP8MI_KERNEL_GEMV;   // M == 1 (the reference's own rule, fp8_mps_native.py:193-210)
    static const int kCandidates[] = {FP8MI_KERNEL_GEMV_MX, FP8MI_KERNEL_SKINNY, FP8MI_KERNEL_GEMM_32x32, FP8MI_KERNEL_GEMM_32x64, FP8MI_KERNEL_GEMM_64x64,
                                      FP8MI_KERNEL_GEMM_64x128, FP8MI_KERNEL_GEMM_128x64, FP8MI_KERNEL_GEMM_128, FP8MI_KERNEL_GEMM_128D,
                                      FP8MI_KERNEL_GEMM_256x128W, FP8MI_KERNEL_GEMM_256W, FP8MI_KERNEL_GEMM_256};
    int best = FP8MI_KERNEL_GENERIC;   // K = 0, unaligned operands, or nothing above takes the shape
    double best_us = 1e300;
    for (int k : kCandidates) {
        double us = predict_us(p, k, cus);
        // no dispatch of any kernel takes less than ~4 us on this runtime (roofline.floor.empty_launch_us): below that the fitted forms only
        // extrapolate (they were fitted on K, N >= 1024), so tiny problems tie there and the FIRST candidate - the smallest kernel that takes
        // the shape - runs
        if (us >= 0.0 && us < kDispatchFloorUs) us = kDispatchFloorUs;
        if (us >= 0.0 && us < best_us) { best_us = us; best = k; }
    }
    return best;
}

}  // namespace fp8mi_dispatch
