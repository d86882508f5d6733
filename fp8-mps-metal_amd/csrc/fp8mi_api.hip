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
#include "fp8mi_dispatch.h"

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

// The kernel the automatic dispatch runs for a problem: the cheapest by the cost model of fp8mi_dispatch.h (host-only arithmetic on shapes,
// strides, alignment and the CU count; exported as fp8mi_choose_kernel, tested on the CPU against measured times).
static int choose_kernel(const MMParams &p) { return fp8mi_dispatch::choose(p, (double)fp8mi_cu_count()); }

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

static MMParams shape_only_params(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int out_dtype, int has_workspace, int split_k)
{
    MMParams p = {};
    p.A = (const uint8_t *)(uintptr_t)0x10000; p.B = (const uint8_t *)(uintptr_t)0x20000; p.C = (void *)(uintptr_t)0x30000;   // aligned, never dereferenced
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.out_dtype = out_dtype;
    p.nan_zero = 1;
    p.split = has_workspace ? split_k : 1;
    p.ws = has_workspace ? (uint8_t *)(uintptr_t)0x40000 : nullptr;
    p.ws_bytes = has_workspace ? fp8mi_scaled_mm_workspace_bytes() : 0;
    return p;
}

int fp8mi_choose_kernel(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int out_dtype, int has_workspace,
                        int split_k)
{
    if (M < 0 || N < 0 || K < 0 || !dtype_ok(out_dtype) || split_k < 0) return FP8MI_E_ENUM;
    return choose_kernel(shape_only_params(M, N, K, lda, ldb, ldc, out_dtype, has_workspace, split_k));
}

double fp8mi_predict_kernel_us(int kernel, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int out_dtype, int has_workspace,
                               int split_k, int compute_units)
{
    if (M < 0 || N < 0 || K < 0 || !dtype_ok(out_dtype) || split_k < 0) return -1.0;
    const MMParams p = shape_only_params(M, N, K, lda, ldb, ldc, out_dtype, has_workspace, split_k);
    return fp8mi_dispatch::predict_us(p, kernel, compute_units > 0 ? (double)compute_units : (double)fp8mi_cu_count());
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
