/*
 * Torch-free use of the C ABI (include/fp8mi.h): a plain C host program that
 * allocates with HIP, calls libfp8mi.so and checks the results against the C
 * oracle (oracle/fp8_oracle.c).  Built and run by tests/test_gpu_c_abi.py:
 *   gcc -D__HIP_PLATFORM_AMD__ tests/c/abi_roundtrip.c oracle/fp8_oracle.c -I/opt/rocm/include -Iinclude \
 *       -Lfp8-mps-metal_amd -lfp8mi -L/opt/rocm/lib -lamdhip64 -lm -o ...
 * Exit code 0 = every check passed.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fp8mi.h"

void fp8o_encode(const float *in, uint8_t *out, size_t n);
void fp8o_dequant_f16_bits(const uint8_t *in, uint16_t *out, size_t n);
void fp8o_scaled_mm_f64(const uint8_t *A, const uint8_t *B, double *C, const float *sa, const float *sb, size_t M,
                        size_t N, size_t K, size_t na, size_t nb);
void fp8o_decode_lut(float *out256);

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_MI(x) do { int r_ = (x); if (r_ != 0) { printf("fp8mi error %d: %s (%s:%d)\n", r_, fp8mi_last_error(), __FILE__, __LINE__); return 3; } } while (0)

static uint32_t rng = 12345u;
static uint32_t next(void) { rng = rng * 1664525u + 1013904223u; return rng >> 8; }

static int run_mm(int M, int K, int N, double tol, int split_k)
{
    uint8_t *A = malloc((size_t)M * K), *B = malloc((size_t)N * K);
    for (size_t i = 0; i < (size_t)M * K; ++i) A[i] = (uint8_t)next();   /* NaN bytes included: reference mode */
    for (size_t i = 0; i < (size_t)N * K; ++i) B[i] = (uint8_t)next();
    float *sa = malloc(4 * M), *sb = malloc(4 * N);
    for (int i = 0; i < M; ++i) sa[i] = 0.005f + (next() % 100) * 1e-4f;
    for (int i = 0; i < N; ++i) sb[i] = 0.005f + (next() % 100) * 1e-4f;
    uint8_t *dA, *dB; float *dC, *dsa, *dsb;
    CHECK_HIP(hipMalloc((void **)&dA, (size_t)M * K)); CHECK_HIP(hipMalloc((void **)&dB, (size_t)N * K));
    CHECK_HIP(hipMalloc((void **)&dC, sizeof(float) * M * N));
    CHECK_HIP(hipMalloc((void **)&dsa, 4 * M)); CHECK_HIP(hipMalloc((void **)&dsb, 4 * N));
    CHECK_HIP(hipMemcpy(dA, A, (size_t)M * K, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(dB, B, (size_t)N * K, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dsa, sa, 4 * M, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(dsb, sb, 4 * N, hipMemcpyHostToDevice));
    void *ws = NULL; int64_t ws_bytes = 0;
    if (split_k != 1) {  /* split-K: the host owns the workspace; its counter block starts out zero */
        ws_bytes = fp8mi_scaled_mm_workspace_bytes();
        CHECK_HIP(hipMalloc(&ws, (size_t)ws_bytes));
        CHECK_HIP(hipMemset(ws, 0, FP8MI_WS_COUNTER_BYTES));
        for (int rep = 0; rep < 2; ++rep)  /* twice: the first launch must leave the counters zero for the second */
            CHECK_MI(fp8mi_scaled_mm_ws(dA, dB, dC, dsa, dsb, NULL, NULL, M, N, K, K, K, N, FP8MI_SCALE_ROW, FP8MI_SCALE_ROW,
                                        FP8MI_F32, 0, FP8MI_NAN_ZERO, FP8MI_KERNEL_AUTO, split_k, ws, ws_bytes, NULL));
    } else {
        CHECK_MI(fp8mi_scaled_mm(dA, dB, dC, dsa, dsb, NULL, NULL, M, N, K, K, K, N, FP8MI_SCALE_ROW, FP8MI_SCALE_ROW, FP8MI_F32, 0,
                                 FP8MI_NAN_ZERO, NULL));
    }
    CHECK_HIP(hipDeviceSynchronize());
    float *C = malloc(sizeof(float) * M * N); double *E = malloc(sizeof(double) * M * N);
    CHECK_HIP(hipMemcpy(C, dC, sizeof(float) * M * N, hipMemcpyDeviceToHost));
    fp8o_scaled_mm_f64(A, B, E, sa, sb, M, N, K, M, N);
    float lut[256]; fp8o_decode_lut(lut);
    double worst = 0;
    for (int m = 0; m < M; ++m)
        for (int n = 0; n < N; ++n) {
            double bound = 0;
            for (int k = 0; k < K; ++k) bound += fabs((double)lut[A[(size_t)m * K + k]] * lut[B[(size_t)n * K + k]]);
            bound *= (double)sa[m] * sb[n];
            double r = fabs(C[(size_t)m * N + n] - E[(size_t)m * N + n]) / (bound + 1e-300);
            if (r > worst) worst = r;
        }
    printf("scaled_mm M=%d K=%d N=%d split_k=%d: max err / sum|ab| = %.3e (tol %.1e)\n", M, K, N, split_k, worst, tol);
    if (ws) hipFree(ws);
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dsa); hipFree(dsb); free(A); free(B); free(C); free(E); free(sa); free(sb);
    return worst <= tol ? 0 : 1;
}

/* K = 0 against a wide N through the workspace entry point with 16-byte "strides" (ADVICE r3): the automatic dispatch must not hand the
   problem to a tile kernel that refuses K = 0 - the result is the empty sum, 0 (no bias here), from the generic kernel */
static int run_k0(void)
{
    const int M = 4, N = 16384;
    float *dC, *dsa, *dsb, one = 1.0f;
    CHECK_HIP(hipMalloc((void **)&dC, sizeof(float) * M * N)); CHECK_HIP(hipMemset(dC, 0x5A, sizeof(float) * M * N));
    CHECK_HIP(hipMalloc((void **)&dsa, 4)); CHECK_HIP(hipMalloc((void **)&dsb, 4));
    CHECK_HIP(hipMemcpy(dsa, &one, 4, hipMemcpyHostToDevice)); CHECK_HIP(hipMemcpy(dsb, &one, 4, hipMemcpyHostToDevice));
    void *ws; int64_t ws_bytes = fp8mi_scaled_mm_workspace_bytes();
    CHECK_HIP(hipMalloc(&ws, (size_t)ws_bytes)); CHECK_HIP(hipMemset(ws, 0, FP8MI_WS_COUNTER_BYTES));
    if (fp8mi_choose_kernel(M, N, 0, 16, 16, N, FP8MI_F32, 1, 0) != FP8MI_KERNEL_GENERIC) { printf("K = 0: AUTO does not pick the generic kernel\n"); return 1; }
    CHECK_MI(fp8mi_scaled_mm_ws(NULL, NULL, dC, dsa, dsb, NULL, NULL, M, N, 0, 16, 16, N, FP8MI_SCALE_TENSOR, FP8MI_SCALE_TENSOR, FP8MI_F32, 0,
                                FP8MI_NAN_ZERO, FP8MI_KERNEL_AUTO, 0, ws, ws_bytes, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float *C = malloc(sizeof(float) * M * N);
    CHECK_HIP(hipMemcpy(C, dC, sizeof(float) * M * N, hipMemcpyDeviceToHost));
    for (int i = 0; i < M * N; ++i) if (C[i] != 0.0f) { printf("K = 0: C[%d] = %g, expected 0\n", i, C[i]); return 1; }
    printf("scaled_mm K=0 M=%d N=%d through fp8mi_scaled_mm_ws: all zero\n", M, N);
    hipFree(dC); hipFree(dsa); hipFree(dsb); hipFree(ws); free(C);
    return 0;
}

int main(void)
{
    printf("libfp8mi version %#x\n", fp8mi_version());
    fp8mi_device_info_t info;
    CHECK_MI(fp8mi_device_info(0, &info));
    printf("device: %s (%s), %d CUs\n", info.name, info.arch, info.compute_units);

    /* decode: all 256 patterns, bit-exact against the oracle */
    uint8_t bytes[256]; uint16_t exp16[256], got16[256];
    for (int i = 0; i < 256; ++i) bytes[i] = (uint8_t)i;
    fp8o_dequant_f16_bits(bytes, exp16, 256);
    uint8_t *dbytes; uint16_t *dh;
    CHECK_HIP(hipMalloc((void **)&dbytes, 256)); CHECK_HIP(hipMalloc((void **)&dh, 512));
    CHECK_HIP(hipMemcpy(dbytes, bytes, 256, hipMemcpyHostToDevice));
    CHECK_MI(fp8mi_dequant(dbytes, dh, NULL, 256, FP8MI_F16, NULL));
    CHECK_HIP(hipMemcpy(got16, dh, 512, hipMemcpyDeviceToHost));
    if (memcmp(exp16, got16, 512)) { printf("decode mismatch\n"); return 1; }
    printf("dequant: 256 patterns bit-exact\n");

    /* encode: 1M values, byte-exact against the oracle */
    const size_t n = 1 << 20;
    float *x = malloc(4 * n); uint8_t *e = malloc(n), *g = malloc(n);
    for (size_t i = 0; i < n; ++i) { float mag = ldexpf(1.0f + (next() % 4096) / 4096.0f, (int)(next() % 26) - 14); x[i] = (next() & 1) ? -mag : mag; }
    x[0] = 0.0f; x[1] = -0.0f; x[2] = INFINITY; x[3] = 448.0f; x[4] = 1.9375f; x[5] = 0.0015f;
    fp8o_encode(x, e, n);
    float *dx; uint8_t *de;
    CHECK_HIP(hipMalloc((void **)&dx, 4 * n)); CHECK_HIP(hipMalloc((void **)&de, n));
    CHECK_HIP(hipMemcpy(dx, x, 4 * n, hipMemcpyHostToDevice));
    CHECK_MI(fp8mi_encode(dx, FP8MI_F32, de, NULL, (int64_t)n, FP8MI_ENC_REFERENCE, NULL));
    CHECK_HIP(hipMemcpy(g, de, n, hipMemcpyDeviceToHost));
    if (memcmp(e, g, n)) { printf("encode mismatch\n"); return 1; }
    printf("encode: %zu values byte-exact\n", n);

    /* argument errors come back as codes + message, no abort */
    if (fp8mi_scaled_mm(NULL, NULL, NULL, NULL, NULL, NULL, NULL, 4, 4, 4, 4, 4, 4, 0, 0, 0, 0, 0, NULL) != FP8MI_E_NULL) return 1;
    printf("error path: \"%s\"\n", fp8mi_last_error());

    int rc = 0;
    rc |= run_mm(1, 2048, 96, 4e-6, 1);      /* GEMV, fp32 VALU */
    rc |= run_mm(4, 1024, 64, 1e-3, 1);      /* skinny, MFMA */
    rc |= run_mm(200, 528, 136, 1e-3, 1);    /* tile GEMM, ragged, K tail */
    rc |= run_mm(3, 100, 7, 4e-6, 1);        /* generic (K % 16 != 0) */
    rc |= run_mm(96, 4096, 200, 1e-3, 0);    /* tile GEMM, split-K chosen by the library, host-owned workspace */
    rc |= run_mm(40, 2064, 130, 1e-3, 3);    /* forced 3 slices, K tail in the last one */
    rc |= run_k0();
    printf(rc ? "FAILED\n" : "C ABI round trip: ok\n");
    return rc;
}
