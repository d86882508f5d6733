// extern "C" surface of libfp8mi.so: argument validation, kernel selection,
// error reporting.  See include/fp8mi.h for the contract of every entry point
// and the reference interface each one replaces.

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <vector>

#include "fp8mi_common.h"

int fp8mi_launch_dequant(const uint8_t *in, void *out, const float *scale, int64_t count, int out_dtype, hipStream_t s);
int fp8mi_launch_encode(const void *in, int in_dtype, uint8_t *out, const float *prescale, int64_t count, int mode,
                        hipStream_t s);
int fp8mi_launch_amax(const void *in, int in_dtype, float *out, int64_t count, hipStream_t s);
int fp8mi_launch_quantize(const void *in, int in_dtype, uint8_t *out, float *scales, int64_t count, int mode,
                          hipStream_t s);

namespace {

thread_local char g_err[256] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_result(int rc, const char *what)
{
    if (rc > 0) return fail(rc, "%s: %s", what, hipGetErrorString((hipError_t)rc));
    if (rc < 0) return fail(rc, "%s: unsupported problem for the selected kernel", what);
    return 0;
}

struct ProfileState {
    std::vector<hipEvent_t> ev;  // start0, stop0, start1, stop1, ...
    int used = 0;
    bool on = false;
};
thread_local ProfileState g_prof;

bool dtype_ok(int d) { return d == FP8MI_F32 || d == FP8MI_F16 || d == FP8MI_BF16; }

}  // namespace

bool fp8mi_next_profile_events(hipEvent_t *start, hipEvent_t *stop)
{
    ProfileState &ps = g_prof;
    if (!ps.on || (size_t)(2 * ps.used + 1) >= ps.ev.size()) return false;
    *start = ps.ev[2 * ps.used];
    *stop = ps.ev[2 * ps.used + 1];
    ++ps.used;
    return true;
}

int fp8mi_cu_count()
{
    static int cache[64];  // 0 = not asked yet; benign race: every thread writes the same value
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cache[dev];
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cache[dev] = n;
    }
    return n;
}

// The kernel the automatic dispatch runs for a problem (host-only arithmetic on shapes, strides, alignment and the CU count;
// exported as fp8mi_choose_kernel and tested on the CPU).  Rules and the measurements behind them:
static int choose_kernel(const MMParams &p)
{
    if (fp8mi_gemv_supported(p)) return FP8MI_KERNEL_GEMV;
    if (p.M >= 2 && p.M <= 8 && p.K > 0 && p.K <= 6144 && fp8mi_gemm_supported(p) && (p.N + 63) / 64 >= fp8mi_cu_count() / 2) {   // (M = 2 since the regret sweep: K=1024 N=16384 5.3 against 9.3 us, K=2048 N=13824 7.3 against 8.8)
        // a few rows against a WIDE, shallow weight matrix: N alone fills the chip with unsplit 32-row tiles, which stream W through the LDS-DMA ring
        // while the few-rows kernel re-reads x per group of weight rows (round 3, tools/time_shape.py: M=4 K=4096 N=14336 12.1 against 14.6 us,
        // M=8 11.9 against 14.8; M=4 K=3072 N=12288 8.8 against 12.1; M=8 K=4096 N=8192 9.8 against 13.0; M=2: equal, stays below); up to N = 8192 the
        // 32x32 tile (twice the workgroups; K=2048 N=8192 M=4: 5.6 against 7.8 us)
        // (beyond one round of 64-column tiles - N > 64 CUs - 64x128 tiles: K=4096 N=28672 M=8 18.4 against 20.2 us)
        if ((p.N + 63) / 64 > fp8mi_cu_count()) return FP8MI_KERNEL_GEMM_64x128;
        return p.N <= 8192 ? FP8MI_KERNEL_GEMM_32x32 : FP8MI_KERNEL_GEMM_32x64;
    }
    const double nk_bytes = (double)p.N * (double)p.K;
    // (2..4 rows: the few-rows kernel keeps N < 5120 - K=N=4096 M=4: 6.1 against 7.4 us, K=14336 N=4096: 12.2 against 13 - but not wider matrices - M=4 K=13824 N=9216: 25.4 against
    //  20.0 us on 32x64 tiles, K=9216 N=8192 18.8 against 15.8, K=2560 N=5120 6.4 against 5.3 - nor K > 16384, which it does not take: M=4 K=28672 N=4096 20.5 against 24.2)
    const bool small_rows = p.M >= 5 || (p.M >= 2 && !fp8mi_gemv_mx_supported(p)) || (p.M >= 3 && p.N >= 5120);   // (M = 2 stays: K=8192 N=5120 8.8 against 13.5 us)
    // (129..256 rows against a small, NARROW matrix - 128x64 tiles on at most a quarter of the CUs: M=160 K=8192 N=1024 8.8 against 14.4 us, M=160 K=2560 N=1536 5.2 against 7.4;
    //  with a wide shallow one the small tiles only multiply the x traffic: M=256 K=1536 N=7168 15.7 against 7.8)
    const bool narrow = p.M <= 256 && nk_bytes <= 12.0 * 1048576.0 && ((p.M + 127) / 128) * ((p.N + 63) / 64) * 4 <= fp8mi_cu_count();
    if (small_rows && (p.M <= 128 || narrow) && p.K >= 1024 && p.ws && p.split != 1 && fp8mi_gemm_supported(p) && nk_bytes >= 1.0 * 1048576.0) {
        // The small-batch regime on the SMALL tile kernels with the automatic K split (round 3; tools/sweep_decode.py on MI355X, six (K, N)
        // from 4096^2 to 14336 x 4096, M = 9 .. 128): 32x64 tiles for M <= 32 (K=14336 N=4096: 12.9-13.2 us against 16.5-17.2 on 64x128 x 8
        // slices; K=12288 N=3072 11.4 against 15.8-16.2; K=N=4096 8.2-8.4 against the skinny kernel's 8.5-10.2), 64x64 tiles for M <= 64
        // (14.3-14.9 against 17.2-17.4) and on to M = 128 while 128x64 tiles would leave half the CUs idle (K=N=4096 M=96: 9.5 against 13.2;
        // K=N=8192 M=96: 19.0 against 22.5) - except M > 96 against K > 8192, where 128x64 x split stays 3-6 % ahead.
        // From M = 5 (the LLM-shape sweep, profiles/r03_llm_shapes.txt: M=8 K=13824 N=5120 14.6 against 26.6 us on the few-rows kernel, K=4096 N=6144 7.7
        // against 11.8; M <= 4 stays there).  A matrix so wide that 64-column tiles need more than one round (N > 64 CUs) takes 64x128 tiles for M <= 64
        // (K=5120 N=27648 M=64: 26.9 against 32.8 us).
        // End of round 3 (tools/sweep_regret.py, profiles/r03_regret.txt): the entry conditions were K >= 2048 and N K >= 12 MiB; on SMALLER weight matrices the
        // 32x32 tile wins just as clearly (M=8 K=7168 N=1536: 7.0 against 11.8 us on the few-rows kernel; M=128 K=3072 N=2048: 5.9 against 9.5 on 128x64;
        // M=12 K=1024 N=7168: 4.3 against 6.0 on the skinny kernel; M=48 K=1536 N=8192: 5.2 against 7.4): now K >= 1024 and N K >= 1 MiB.
        const int64_t cus = fp8mi_cu_count(), t64 = ((p.M + 127) / 128) * ((p.N + 63) / 64);
        if (p.M <= 64 && (p.N + 63) / 64 > cus) return FP8MI_KERNEL_GEMM_64x128;
        if (nk_bytes <= 12.0 * 1048576.0 || p.K <= 4096) {
            // A small matrix or a shallow K (end of round 3, tools/sweep_regret.py): the LARGEST of the three small tiles whose grid fills the chip (75-100 % of the CUs)
            // - M=64 K=1024 N=12288: 64x64 = 192 tiles 4.9 us, 32x64 = 384 tiles 7.2; M=48 K=3072 N=6144: 32x64 = 192 tiles 8.3, 64x64 = 96 tiles x 2 slices 10.1;
            // M=128 K=3072 N=2048: 32x32 = 256 tiles 5.9, 64x64 = 64 tiles 7.5 - and when none does, the SMALLEST whose grid can be split (at most half the CUs; against a
            // shallow K an unsplit grid on 50-75 % of the CUs still beats the split: M=16 K=2048 N=5120, 160 tiles of 32x32 4.9 us, 80 of 32x64 x 2 slices 6.5;
            // M=160 K=8192 N=1024: 32x32 = 160 tiles 11.1, 32x64 = 80 tiles x 3 slices 8.8)
            const int64_t t3 = ((p.M + 63) / 64) * ((p.N + 63) / 64), t2 = ((p.M + 31) / 32) * ((p.N + 63) / 64), t1 = ((p.M + 31) / 32) * ((p.N + 31) / 32);
            auto fills = [&](int64_t t, int quarters) { return t * 4 >= cus * quarters && t <= cus; };   // (one round: M=96 K=4096 N=14336, 448 tiles of 64x64, 19.6 us against 16.0 on 224 of 128x64)
            auto splits = [&](int64_t t) { return t <= cus / 2 || (p.K <= 4096 && t <= 2 * cus); };
            // (against a shallow K half a round unsplit still beats more than one round of a smaller tile: M=48 K=4096 N=10240, 160 tiles of 64x64 11.9 us, 320 of 32x64 17.2)
            for (int quarters = 3; quarters >= (p.K <= 4096 ? 2 : 3); --quarters) {
                if (p.M > 32 && fills(t3, quarters)) return FP8MI_KERNEL_GEMM_64x64;
                if (fills(t2, quarters)) return FP8MI_KERNEL_GEMM_32x64;
                if (fills(t1, quarters) && p.K <= 8192) return FP8MI_KERNEL_GEMM_32x32;   // (against a deeper K the smallest tile's doubled x traffic loses to a split: M=192 K=12288 N=1024 16.3 against 12.3 us)
            }
            // (deep K: the larger tile first while its slices stay at least 2 KiB deep - M=192 K=12288 N=1024: 64x64 x 5 slices 12.3 us, 32x64 x 2 15.1; but M=64 K=9216 N=1024:
            //  16 tiles of 64x64 x 16 slices 10.0, 64 of 32x32 x 4 7.6)
            auto deep_enough = [&](int64_t t) { return t * p.K >= cus * 2048; };
            if (p.K > 8192 && p.M > 32 && splits(t3) && deep_enough(t3)) return FP8MI_KERNEL_GEMM_64x64;
            if (p.K > 8192 && splits(t2) && deep_enough(t2)) return FP8MI_KERNEL_GEMM_32x64;
            if (splits(t1)) return FP8MI_KERNEL_GEMM_32x32;
            if (splits(t2)) return FP8MI_KERNEL_GEMM_32x64;
            if (p.M > 128) return FP8MI_KERNEL_GEMM_64x64;
        }
        // ... and 32x32 tiles where K and N stay within 8192: twice the tiles = half the K slices (none at N = 8192), the partial exchange being what the
        // regime pays for (K=N=8192 M=32: 14.7 against 18.0 us; K=N=4096: 7.2 against 8.1); against a deeper K their doubled x traffic loses (K=12288 N=3072: 12.3 against 11.6)
        if (p.M <= 32) return ((p.N <= 8192 && p.K <= 8192) || p.N <= 2048) ? FP8MI_KERNEL_GEMM_32x32 : FP8MI_KERNEL_GEMM_32x64;   // (a narrow N at any K: M=16 K=14336 N=1536 9.1 against 10.5 us)
        if (p.M <= 64 && (double)p.N * (double)p.K <= 16.0 * 1048576.0 && ((p.M + 31) / 32) * ((p.N + 31) / 32) <= 2 * cus) return FP8MI_KERNEL_GEMM_32x32;   // (K=N=4096, M=48-64: 7.6 against 9.4 us on 64x64; at most two rounds: M=64 K=1024 N=16384, 1024 tiles, 7.9 against 5.5)
        // (33..64 rows against N = 8192-ish: two rows of 32x64 tiles are one whole round with NO K split, i.e. no partial exchange, where 64x64 tiles need
        //  2 slices - K=N=8192 M=33 / 48 / 64: 15.7 / 17.4 / 18.0 against 19.3 / 19.2 / 19.1 us, K=4096 9.4-10.5 against 10.4-10.8; at N = 7936, 248 tiles, 64x64 x 2 is 6 % ahead again and stays)
        if (p.M <= 64 && p.K <= 8192 && 2 * ((p.N + 63) / 64) <= cus && 2 * ((p.N + 63) / 64) > cus - 4) return FP8MI_KERNEL_GEMM_32x64;
        if (p.M <= 64) {
            // (64x64 tiles unsplit on 50-75 % of the CUs against a deep K: 64x128 tiles with the K split instead - M=64 K=14336 N=9216: 27.8 against 35.7 us)
            const int64_t t = (p.N + 63) / 64;
            if (p.K >= 8192 && t > cus / 2 && t * 4 <= cus * 3) return FP8MI_KERNEL_GEMM_64x128;
            return FP8MI_KERNEL_GEMM_64x64;
        }
        // (65..128 rows: 64x64 tiles either fill at most half the chip - then the K split does - or most of it; in between they run unsplit on ~60 % of the CUs:
        //  M=96 K=28672 N=5120, 160 tiles: 57.5 us against 40.2 on 128x64 x 3 slices)
        const int64_t t6464 = ((p.M + 63) / 64) * ((p.N + 63) / 64);
        const bool underfilled = p.K >= 8192 && t6464 > cus / 2 && t6464 * 4 < cus * 3;
        if (t64 <= cus / 2 && !(p.M > 96 && p.K > 8192 && p.N >= 4096) && !underfilled) return FP8MI_KERNEL_GEMM_64x64;   // (a narrow N keeps 64x64 at M = 128 too: K=13824 N=1536 12.9 against 15.3 us)
    }
    if (fp8mi_gemv_mx_supported(p)) {
        // 2..8 rows of x on the vec-mat's weight-streaming structure (tools/check_gemv_mx.py time, MI355X): ahead of
        // the skinny and the split-K tile kernel for M <= 4 everywhere measured (K = N = 4096: 5.6 / 6.3 vs 7.8 us;
        // K = 14336, N = 4096: 12.2 vs 16-17 us) and for 5 <= M <= 8 once K > 4096 on matrices the tile kernel
        // cannot fill the chip with (K = 14336, N = 4096: 14.3 vs 17.4 us; K = N = 8192: 15.1 vs 17.9 us)
        const int64_t t64 = (p.N + 63) / 64, cus = fp8mi_cu_count();
        if (p.M <= 4 || (p.K > 4096 && t64 < (3 * cus) / 4)) return FP8MI_KERNEL_GEMV_MX;
    }
    if (p.M >= 2 && p.M <= 48 && fp8mi_skinny_supported(p)) {
        // measured (tools/sweep_small_m.py): on small weight matrices the weight-streaming skinny kernel wins up
        // to M = 48 (K = N = 4096: 8.5-11.4 vs 11.4-12.6 us); on large ones (N*K >= 24 MiB) the split-K tile
        // kernel wins from M = 2 (K = 4096, N = 14336: 13.7 vs 25.4 us; M = 32, K = 14336, N = 4096: 17.7 vs
        // 28.4 us) - it needs the workspace
        const bool big = (double)p.N * (double)p.K >= 24.0 * 1048576.0;
        // ... and when N alone yields >= 192 tiles of 128x64 the tile kernel needs no split to fill the chip
        // (K = 4096, N = 14336: 14-15 us for every M <= 64, skinny 20-36 us: x is re-read by every 16-row workgroup)
        // (from 128 tiles on while K <= 4096: K = 4096, N = 8192: 14-15 vs 16-19 us; at K = 8192 the 128 busy CUs lose)
        const int64_t t64 = (p.N + 63) / 64, cus = fp8mi_cu_count();
        const bool wide = t64 >= (3 * cus) / 4 || (t64 >= cus / 2 && p.K <= 4096);
        if (!(fp8mi_gemm_supported(p) && ((p.ws && p.split != 1 && big) || wide))) return FP8MI_KERNEL_SKINNY;
    }
    if (p.K > 0 && fp8mi_gemm_supported(p)) return fp8mi_choose_gemm_variant(p);
    return FP8MI_KERNEL_GENERIC;
}

extern "C" {

int fp8mi_profile_begin(int max_launches)
{
    if (max_launches <= 0) return fail(FP8MI_E_SHAPE, "fp8mi_profile_begin: max_launches must be positive");
    ProfileState &ps = g_prof;
    if (ps.on) return fail(FP8MI_E_UNSUPPORTED, "fp8mi_profile_begin: a profile is already open on this thread");
    while (ps.ev.size() < (size_t)max_launches * 2) {
        hipEvent_t e;
        hipError_t rc = hipEventCreate(&e);
        if (rc != hipSuccess) return fail((int)rc, "hipEventCreate: %s", hipGetErrorString(rc));
        ps.ev.push_back(e);
    }
    ps.used = 0;
    ps.on = true;
    return 0;
}

int fp8mi_profile_end(float *ms_out, int cap)
{
    ProfileState &ps = g_prof;
    if (!ps.on) return fail(FP8MI_E_UNSUPPORTED, "fp8mi_profile_end: no profile open on this thread");
    ps.on = false;
    const int n = ps.used;
    if (n > 0) {
        hipError_t rc = hipEventSynchronize(ps.ev[2 * (n - 1) + 1]);
        if (rc != hipSuccess) return fail((int)rc, "hipEventSynchronize: %s", hipGetErrorString(rc));
    }
    for (int i = 0; i < n && i < cap && ms_out; ++i) {
        float ms = 0.0f;
        hipError_t rc = hipEventElapsedTime(&ms, ps.ev[2 * i], ps.ev[2 * i + 1]);
        if (rc != hipSuccess) return fail((int)rc, "hipEventElapsedTime: %s", hipGetErrorString(rc));
        ms_out[i] = ms;
    }
    return n;
}

int fp8mi_version(void) { return FP8MI_VERSION; }

const char *fp8mi_last_error(void) { return g_err; }

int fp8mi_device_info(int device, fp8mi_device_info_t *out)
{
    if (!out) return fail(FP8MI_E_NULL, "fp8mi_device_info: out is NULL");
    hipDeviceProp_t pr;
    hipError_t e = hipGetDeviceProperties(&pr, device);
    if (e != hipSuccess) return fail((int)e, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    memset(out, 0, sizeof(*out));
    out->compute_units = pr.multiProcessorCount;
    out->clock_khz = pr.clockRate;
    out->memory_clock_khz = pr.memoryClockRate;
    out->memory_bus_bits = pr.memoryBusWidth;
    out->l2_bytes = pr.l2CacheSize;
    out->lds_bytes_per_cu = (int)pr.maxSharedMemoryPerMultiProcessor;
    out->wavefront_size = pr.warpSize;
    out->total_memory = (int64_t)pr.totalGlobalMem;
    strncpy(out->arch, pr.gcnArchName, sizeof(out->arch) - 1);
    strncpy(out->name, pr.name, sizeof(out->name) - 1);
    return 0;
}

int64_t fp8mi_scaled_mm_workspace_bytes(void)
{
    // counters + room for the fp32 partial tiles of every split the library takes on its own, on any of the split-capable ring tiles
    // (128x64 / 64x128: 32 KiB per workgroup; 64x64 16 KiB, 32x64 8 KiB, 32x32 4 KiB; 128x128 64 KiB): tiles x slices stays within one
    // workgroup per CU, 256 x 32 KiB at most; twice that leaves room for forced splits (split_k > 0).  A split whose partials do not fit
    // the caller's workspace is clamped to the largest slice count that does (resolve_split, fp8mi_gemm_epi.h).
    return (int64_t)FP8MI_WS_COUNTER_BYTES + 2 * 256 * (int64_t)(128 * 64 * 4);
}

int fp8mi_workspace_reset(void *workspace, int64_t workspace_bytes, void *stream)
{
    if (!workspace) return fail(FP8MI_E_NULL, "fp8mi_workspace_reset: workspace is NULL");
    if (workspace_bytes < FP8MI_WS_COUNTER_BYTES) return fail(FP8MI_E_SHAPE, "fp8mi_workspace_reset: workspace smaller than the counter block");
    hipError_t e = hipMemsetAsync(workspace, 0, FP8MI_WS_COUNTER_BYTES, (hipStream_t)stream);
    if (e != hipSuccess) return fail((int)e, "hipMemsetAsync: %s", hipGetErrorString(e));
    return 0;
}

int fp8mi_scaled_mm_ex(const uint8_t *A, const uint8_t *B_nk, void *C, const float *scale_a, const float *scale_b,
                       const void *bias, const float *scale_result, int64_t M, int64_t N, int64_t K, int64_t lda,
                       int64_t ldb, int64_t ldc, int scale_a_mode, int scale_b_mode, int out_dtype, int bias_dtype,
                       int nan_mode, int kernel, void *stream)
{
    return fp8mi_scaled_mm_ws(A, B_nk, C, scale_a, scale_b, bias, scale_result, M, N, K, lda, ldb, ldc, scale_a_mode,
                              scale_b_mode, out_dtype, bias_dtype, nan_mode, kernel, 1, nullptr, 0, stream);
}

int fp8mi_scaled_mm_ws(const uint8_t *A, const uint8_t *B_nk, void *C, const float *scale_a, const float *scale_b,
                       const void *bias, const float *scale_result, int64_t M, int64_t N, int64_t K, int64_t lda,
                       int64_t ldb, int64_t ldc, int scale_a_mode, int scale_b_mode, int out_dtype, int bias_dtype,
                       int nan_mode, int kernel, int split_k, void *workspace, int64_t workspace_bytes, void *stream)
{
    if (M < 0 || N < 0 || K < 0) return fail(FP8MI_E_SHAPE, "fp8mi_scaled_mm: negative dimension (M=%lld N=%lld K=%lld)",
                                              (long long)M, (long long)N, (long long)K);
    if (M == 0 || N == 0) return 0;
    if (!C || !scale_a || !scale_b) return fail(FP8MI_E_NULL, "fp8mi_scaled_mm: C / scale_a / scale_b must not be NULL");
    if (K > 0 && (!A || !B_nk)) return fail(FP8MI_E_NULL, "fp8mi_scaled_mm: A / B must not be NULL when K > 0");
    if (lda < K || ldb < K || ldc < N)
        return fail(FP8MI_E_SHAPE, "fp8mi_scaled_mm: leading dimension too small (lda=%lld ldb=%lld ldc=%lld)",
                    (long long)lda, (long long)ldb, (long long)ldc);
    const int transposed = (bias_dtype & FP8MI_EPILOGUE_TRANSPOSED) ? 1 : 0;
    bias_dtype &= ~FP8MI_EPILOGUE_TRANSPOSED;
    if (!dtype_ok(out_dtype) || (bias && !dtype_ok(bias_dtype)))
        return fail(FP8MI_E_ENUM, "fp8mi_scaled_mm: unknown out_dtype / bias_dtype");
    if ((scale_a_mode | 1) != 1 || (scale_b_mode | 1) != 1 || (nan_mode | 1) != 1)
        return fail(FP8MI_E_ENUM, "fp8mi_scaled_mm: unknown scale mode / nan mode");

    MMParams p;
    p.A = A; p.B = B_nk; p.C = C;
    p.scale_a = scale_a; p.scale_b = scale_b; p.bias = bias; p.scale_result = scale_result;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.sa_row = scale_a_mode; p.sb_row = scale_b_mode;
    p.out_dtype = out_dtype; p.bias_dtype = bias_dtype; p.transposed = transposed;
    p.nan_zero = nan_mode == FP8MI_NAN_ZERO;
    p.debug = 0;
#ifdef FP8MI_DIAG
    if (const char *e = getenv("FP8MI_DEBUG")) p.debug = atoi(e);   // diagnostic library only: timing-only ablation bits
#endif
    if (split_k < 0) return fail(FP8MI_E_ENUM, "fp8mi_scaled_mm_ws: split_k must be >= 0");
    if (workspace && ((((uintptr_t)workspace) & 15u) != 0 || workspace_bytes < FP8MI_WS_COUNTER_BYTES)) {
        workspace = nullptr;  // unusable: behave as if none was given
    }
    p.split = workspace ? split_k : 1;
    p.ws = (uint8_t *)workspace;
    p.ws_bytes = workspace ? workspace_bytes : 0;
    hipStream_t s = (hipStream_t)stream;

    if (kernel == FP8MI_KERNEL_AUTO) {
        kernel = choose_kernel(p);   // (every id it returns passes its own envelope check below)
        // The generic kernel is the device path of last resort (one wave per OUTPUT ELEMENT, byte loads): correct for every problem the reference
        // accepts (fp8_mps_native.py:55-60 only asks for contiguity), and orders of magnitude slower than the tile kernels.  A large problem that
        // lands on it - K, lda or ldb not a multiple of 16, or a base pointer that is not 16-byte aligned (a sliced weight view) - says so ONCE per
        // process on stderr (FP8MI_QUIET=1 silences it); the Python op layer pads such operands into aligned copies instead (fp8_mi355x_native.py).
        if (kernel == FP8MI_KERNEL_GENERIC && K > 0 && ((double)N * (double)K >= 1048576.0 || (double)M * (double)K >= 1048576.0)) {
            static std::atomic<bool> told{false};
            if (!told.exchange(true)) {
                const char *q = getenv("FP8MI_QUIET");
                if (!(q && q[0] == '1'))
                    fprintf(stderr, "[fp8mi] note: M=%lld N=%lld K=%lld (lda=%lld ldb=%lld, A %% 16 = %d, B %% 16 = %d) runs on the GENERIC kernel: the MFMA "
                                    "kernels need K, lda and ldb to be multiples of 16 and 16-byte aligned operands. Pad K with zero bytes / align the rows "
                                    "(fp8_mi355x_native.fp8_scaled_mm does so by itself). This note is printed once.\n",
                            (long long)M, (long long)N, (long long)K, (long long)lda, (long long)ldb, (int)((uintptr_t)A & 15), (int)((uintptr_t)B_nk & 15));
            }
        }
    }
    switch (kernel) {
    case FP8MI_KERNEL_GEMV:
        if (!fp8mi_gemv_supported(p)) return fail(FP8MI_E_UNSUPPORTED, "gemv kernel needs M == 1, K %% 16 == 0, 16-byte aligned rows");
        return hip_result(fp8mi_launch_gemv(p, false, s), "gemv");
    case FP8MI_KERNEL_GEMV_MX:
        if (!fp8mi_gemv_mx_supported(p)) return fail(FP8MI_E_UNSUPPORTED, "few-rows kernel needs 2 <= M <= 8, K <= 16384, K %% 16 == 0, 16-byte aligned rows");
        return hip_result(fp8mi_launch_gemv_mx(p, s), "gemv-mx");
    case FP8MI_KERNEL_GEMV_FP32:
        if (!fp8mi_gemv_supported(p)) return fail(FP8MI_E_UNSUPPORTED, "gemv kernel needs M == 1, K %% 16 == 0, 16-byte aligned rows");
        return hip_result(fp8mi_launch_gemv(p, true, s), "gemv-fp32");
    case FP8MI_KERNEL_SKINNY:
        if (!fp8mi_skinny_supported(p)) return fail(FP8MI_E_UNSUPPORTED, "skinny kernel needs 1 <= M <= 64, K %% 16 == 0, 16-byte aligned rows");
        return hip_result(fp8mi_launch_skinny(p, s), "skinny");
    case FP8MI_KERNEL_GEMM_128:
    case FP8MI_KERNEL_GEMM_128x64:
    case FP8MI_KERNEL_GEMM_256:
    case FP8MI_KERNEL_GEMM_64x128:
    case FP8MI_KERNEL_GEMM_64x64:
    case FP8MI_KERNEL_GEMM_32x64:
    case FP8MI_KERNEL_GEMM_32x32:
    case FP8MI_KERNEL_GEMM_128D:
        if (K <= 0 || !fp8mi_gemm_supported(p)) return fail(FP8MI_E_UNSUPPORTED, "MFMA gemm kernel needs K > 0, K %% 16 == 0 and 16-byte aligned rows");
        return hip_result(fp8mi_launch_gemm(p, kernel, s), "gemm");
    case FP8MI_KERNEL_GEMM_256W:
        if (!fp8mi_gemm256_supported(p)) return fail(FP8MI_E_UNSUPPORTED, "256x256 one-wave-per-SIMD kernel needs K >= 256 (> 256 with a K tail), N a multiple of 16 bytes of output, 16-byte aligned rows, no split-K");
        return hip_result(fp8mi_launch_gemm256(p, 0, s), "gemm256");
    case FP8MI_KERNEL_GEMM_256x128W:
        if (!fp8mi_gemm256_supported(p)) return fail(FP8MI_E_UNSUPPORTED, "256x128 one-wave-per-SIMD kernel needs K >= 256 (> 256 with a K tail), N a multiple of 16 bytes of output, 16-byte aligned rows, no split-K");
        return hip_result(fp8mi_launch_gemm256(p, 1000, s), "gemm256x128");
    case FP8MI_KERNEL_GENERIC:
        return hip_result(fp8mi_launch_generic(p, s), "generic");
    default:
#ifdef FP8MI_DIAG
        if (kernel >= 80 && kernel <= 119 && fp8mi_gemm256_supported(p)) return hip_result(fp8mi_launch_gemm256(p, kernel - 80, s), "gemm256-variant");
        if (kernel >= 190 && kernel <= 199 && fp8mi_gemm256_supported(p)) return hip_result(fp8mi_launch_gemm256(p, 1000 + kernel - 190, s), "gemm256x128-variant");
        if (kernel >= 70 && kernel <= 77 && fp8mi_gemv_mx_supported(p)) return hip_result(fp8mi_launch_gemv_mx_variant(p, kernel, s), "gemv-mx-variant");
        if (((kernel >= 40 && kernel <= 69) || (kernel >= 160 && kernel <= 189)) && fp8mi_gemv_supported(p)) return hip_result(fp8mi_launch_gemv_variant(p, kernel, s), "gemv-variant");
#endif
#ifdef FP8MI_DIAG  // diagnostic library only: schedule variants of the ring kernel (7..13, 30..37), the producer / consumer kernel
                   // (15..24) and its timing-only ablations (201..207)
        if (K > 0 && fp8mi_gemm_supported(p)) {
            if ((kernel >= 7 && kernel <= 13) || (kernel >= 30 && kernel <= 39) || (kernel >= 120 && kernel <= 159)) return hip_result(fp8mi_launch_gemm(p, kernel, s), "gemm-variant");
            if ((kernel >= 15 && kernel <= 29) || (kernel >= 200 && kernel < 220)) return hip_result(fp8mi_launch_gemm_pc(p, kernel, s), "gemm-pc");
        }
#endif
        return fail(FP8MI_E_ENUM, "fp8mi_scaled_mm_ex: unknown kernel id %d", kernel);
    }
}

int fp8mi_choose_kernel(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int out_dtype, int has_workspace,
                        int split_k)
{
    if (M < 0 || N < 0 || K < 0 || !dtype_ok(out_dtype) || split_k < 0) return FP8MI_E_ENUM;
    MMParams p = {};
    p.A = (const uint8_t *)(uintptr_t)0x10000; p.B = (const uint8_t *)(uintptr_t)0x20000; p.C = (void *)(uintptr_t)0x30000;   // aligned, never dereferenced
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.out_dtype = out_dtype;
    p.nan_zero = 1;
    p.split = has_workspace ? split_k : 1;
    p.ws = has_workspace ? (uint8_t *)(uintptr_t)0x40000 : nullptr;
    p.ws_bytes = has_workspace ? (int64_t)1 << 30 : 0;
    return choose_kernel(p);
}

int fp8mi_scaled_mm(const uint8_t *A, const uint8_t *B_nk, void *C, const float *scale_a, const float *scale_b,
                    const void *bias, const float *scale_result, int64_t M, int64_t N, int64_t K, int64_t lda,
                    int64_t ldb, int64_t ldc, int scale_a_mode, int scale_b_mode, int out_dtype, int bias_dtype,
                    int nan_mode, void *stream)
{
    return fp8mi_scaled_mm_ex(A, B_nk, C, scale_a, scale_b, bias, scale_result, M, N, K, lda, ldb, ldc, scale_a_mode,
                              scale_b_mode, out_dtype, bias_dtype, nan_mode, FP8MI_KERNEL_AUTO, stream);
}

int fp8mi_dequant(const uint8_t *in, void *out, const float *scale, int64_t count, int out_dtype, void *stream)
{
    if (count < 0) return fail(FP8MI_E_SHAPE, "fp8mi_dequant: negative count");
    if (count == 0) return 0;
    if (!in || !out) return fail(FP8MI_E_NULL, "fp8mi_dequant: in / out must not be NULL");
    if (!dtype_ok(out_dtype)) return fail(FP8MI_E_ENUM, "fp8mi_dequant: unknown out_dtype %d", out_dtype);
    return hip_result(fp8mi_launch_dequant(in, out, scale, count, out_dtype, (hipStream_t)stream), "dequant");
}

int fp8mi_dequant_f16(const uint8_t *in, void *out, const float *scale_or_null, int64_t count, int out_dtype, void *stream)
{
    return fp8mi_dequant(in, out, scale_or_null, count, out_dtype, stream);   // SURVEY.md 8(b)'s name for the same entry point
}

int fp8mi_encode(const void *in, int in_dtype, uint8_t *out, const float *prescale, int64_t count, int encode_mode,
                 void *stream)
{
    if (count < 0) return fail(FP8MI_E_SHAPE, "fp8mi_encode: negative count");
    if (count == 0) return 0;
    if (!in || !out) return fail(FP8MI_E_NULL, "fp8mi_encode: in / out must not be NULL");
    if (!dtype_ok(in_dtype)) return fail(FP8MI_E_ENUM, "fp8mi_encode: unknown in_dtype %d", in_dtype);
    if ((encode_mode | 1) != 1) return fail(FP8MI_E_ENUM, "fp8mi_encode: unknown encode_mode %d", encode_mode);
    return hip_result(fp8mi_launch_encode(in, in_dtype, out, prescale, count, encode_mode, (hipStream_t)stream), "encode");
}

int fp8mi_amax(const void *in, int in_dtype, float *out, int64_t count, void *stream)
{
    if (count < 0) return fail(FP8MI_E_SHAPE, "fp8mi_amax: negative count");
    if (!out || (count > 0 && !in)) return fail(FP8MI_E_NULL, "fp8mi_amax: in / out must not be NULL");
    if (!dtype_ok(in_dtype)) return fail(FP8MI_E_ENUM, "fp8mi_amax: unknown in_dtype %d", in_dtype);
    return hip_result(fp8mi_launch_amax(in, in_dtype, out, count, (hipStream_t)stream), "amax");
}

int fp8mi_quantize(const void *in, int in_dtype, uint8_t *out, float *scales, int64_t count, int encode_mode,
                   void *stream)
{
    if (count < 0) return fail(FP8MI_E_SHAPE, "fp8mi_quantize: negative count");
    if (!scales || (count > 0 && (!in || !out))) return fail(FP8MI_E_NULL, "fp8mi_quantize: NULL pointer");
    if (!dtype_ok(in_dtype)) return fail(FP8MI_E_ENUM, "fp8mi_quantize: unknown in_dtype %d", in_dtype);
    if ((encode_mode | 1) != 1) return fail(FP8MI_E_ENUM, "fp8mi_quantize: unknown encode_mode %d", encode_mode);
    return hip_result(fp8mi_launch_quantize(in, in_dtype, out, scales, count, encode_mode, (hipStream_t)stream), "quantize");
}

}  // extern "C"
