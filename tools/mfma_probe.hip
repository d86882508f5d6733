// Probe: numerical behaviour of the gfx950 fp8 MFMA instructions (one wave).
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// Measures, against a double reference on the host:
//   * v_mfma_scale_f32_16x16x128_f8f6f4 (scale 2^0) - the double-rate path
//   * 4 x v_mfma_f32_16x16x32_fp8_fp8 - the legacy path
// on random bytes and on "one big product + 127 small products" patterns that
// expose how far below the largest product a term can sit before it is lost.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k_scaled(const i32x8 *a, const i32x8 *b, f32x4 *c, int reps)
{
    int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r)
        acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l + 64 * r], b[l + 64 * r], acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    c[l] = acc;
}
__global__ void k_legacy(const i32x8 *a, const i32x8 *b, f32x4 *c, int reps)
{
    int l = threadIdx.x;
    f32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        i32x8 x = a[l + 64 * r], y = b[l + 64 * r];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            long xa = ((long)(unsigned)x[2 * q + 1] << 32) | (unsigned)x[2 * q];
            long yb = ((long)(unsigned)y[2 * q + 1] << 32) | (unsigned)y[2 * q];
            acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(xa, yb, acc, 0, 0, 0);
        }
    }
    c[l] = acc;
}

static double dec(uint8_t b)
{
    if ((b & 0x7F) == 0x7F) return NAN;
    int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    double v = e == 0 ? m / 8.0 * pow(2, -6) : (1 + m / 8.0) * pow(2, e - 7);
    return s ? -v : v;
}

static void run(const char *name, uint8_t *A, uint8_t *B, int reps, int show)
{
    size_t n = (size_t)64 * 32 * reps;
    uint8_t *dA, *dB; float *dC;
    hipMalloc(&dA, n); hipMalloc(&dB, n); hipMalloc(&dC, 64 * 16);
    hipMemcpy(dA, A, n, hipMemcpyHostToDevice); hipMemcpy(dB, B, n, hipMemcpyHostToDevice);
    float C[2][256];
    hipLaunchKernelGGL(k_scaled, 1, 64, 0, 0, (const i32x8 *)dA, (const i32x8 *)dB, (f32x4 *)dC, reps);
    hipMemcpy(C[0], dC, 1024, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k_legacy, 1, 64, 0, 0, (const i32x8 *)dA, (const i32x8 *)dB, (f32x4 *)dC, reps);
    hipMemcpy(C[1], dC, 1024, hipMemcpyDeviceToHost);
    double worst[2] = {0, 0}, worstf32 = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double ex = 0, bd = 0; float f32 = 0;
            for (int r = 0; r < reps; ++r)
                for (int g = 0; g < 4; ++g)
                    for (int t = 0; t < 32; ++t) {
                        double p = dec(A[((size_t)r * 64 + 16 * g + i) * 32 + t]) * dec(B[((size_t)r * 64 + 16 * g + j) * 32 + t]);
                        ex += p; bd += fabs(p); f32 += (float)p;
                    }
            int lane = (i / 4) * 16 + j, reg = i % 4;
            for (int v = 0; v < 2; ++v) {
                double e = fabs(C[v][lane * 4 + reg] - ex) / (bd + 1e-300);
                if (e > worst[v]) worst[v] = e;
            }
            double e = fabs(f32 - ex) / (bd + 1e-300);
            if (e > worstf32) worstf32 = e;
            if (show && i == 0 && j == 0) printf("    D[0][0]: exact %.10g  scaled %.10g  legacy %.10g  seq-f32 %.10g\n", ex, C[0][0], C[1][0], f32);
        }
    printf("%-44s K=%5d  max|err|/sum|ab|: scaled %.3e  legacy %.3e  (sequential f32 %.3e)\n", name, 128 * reps, worst[0], worst[1], worstf32);
    hipFree(dA); hipFree(dB); hipFree(dC);
}

int main()
{
    srand(1234);
    for (int reps : {1, 8, 32, 128}) {
        size_t n = (size_t)64 * 32 * reps;
        uint8_t *A = (uint8_t *)malloc(n), *B = (uint8_t *)malloc(n);
        for (size_t i = 0; i < n; ++i) { A[i] = rand() & 0xFF; if ((A[i] & 0x7F) == 0x7F) A[i] ^= 1; B[i] = rand() & 0xFF; if ((B[i] & 0x7F) == 0x7F) B[i] ^= 1; }
        run("uniform random bytes", A, B, reps, 0);
        for (size_t i = 0; i < n; ++i) { A[i] = 0x28 + (rand() % 0x20); B[i] = 0x28 + (rand() % 0x20) + ((rand() & 1) << 7); }
        run("narrow-range values (|x| in [0.25,4))", A, B, reps, 0);
        free(A); free(B);
    }
    // one big product (256 * 1) + 127 products of 2^-s: how small can they be and still count?
    for (int s = 2; s <= 26; s += 2) {
        uint8_t A[64 * 32], B[64 * 32];
        // small = 2^-(s): a = 2^-(s/2 ... ) choose a = 2^-ea, b = 2^-eb with ea + eb = s, both in [-6, 8] exponent range
        int ea = s / 2, eb = s - ea;           // a = 2^-ea, b = 2^-eb ; need ea,eb <= 6 for normals, use subnormals beyond
        auto enc_pow2 = [](int e) -> uint8_t {  // 2^-e, e in [0, 9]
            if (e <= 6) return (uint8_t)((7 - e) << 3);
            return (uint8_t)(1 << (9 - e));     // subnormal: 2^-7 = 0x04, 2^-8 = 0x02, 2^-9 = 0x01
        };
        if (ea > 9 || eb > 9) { // shift the big one up instead: big = 2^8 * 2^8
            break;
        }
        for (int i = 0; i < 64 * 32; ++i) { A[i] = enc_pow2(ea); B[i] = enc_pow2(eb); }
        for (int l = 0; l < 16; ++l) { A[l * 32] = 0x78; B[l * 32] = 0x38; }  // g = 0, t = 0: 256 * 1
        char nm[64]; snprintf(nm, sizeof nm, "256 + 127 x 2^-%d", s);
        run(nm, A, B, 1, 1);
    }
    // same with the big product at 2^16 (448-scale operands): 2^8 * 2^8
    for (int s = 0; s <= 18; s += 2) {
        uint8_t A[64 * 32], B[64 * 32];
        auto enc_pow2 = [](int e) -> uint8_t { if (e <= 6) return (uint8_t)((7 - e) << 3); return (uint8_t)(1 << (9 - e)); };
        int ea = s / 2, eb = s - ea;
        for (int i = 0; i < 64 * 32; ++i) { A[i] = enc_pow2(ea); B[i] = enc_pow2(eb); }
        for (int l = 0; l < 16; ++l) { A[l * 32] = 0x78; B[l * 32] = 0x78; }
        char nm[64]; snprintf(nm, sizeof nm, "65536 + 127 x 2^-%d", s);
        run(nm, A, B, 1, 1);
    }
    return 0;
}
