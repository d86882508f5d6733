/*
 * CPU oracle (plain C) for the FP8 e4m3fn scaled-matmul + cast path.
 *
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py - never by the product library.
 *
 * Restates the reference's four Metal kernels (audiohacking/fp8-mps-metal);
 * each function cites the reference lines it follows.  Pinned against the
 * golden vectors in tests/golden/ (generated from the reference's executable
 * spec, test_fp8_correctness.py:22-106) by tests/test_oracle.py.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* fp8_matmul.metal:19-40 - byte -> float; NaN patterns -> +0.0 (:21), sign last (:39) */
static float decode1(uint8_t b)
{
    if ((b & 0x7F) == 0x7F) return 0.0f;
    unsigned sign = (b >> 7) & 1u, e = (b >> 3) & 0xFu, m = b & 7u;
    float v;
    if (e == 0) v = (float)m / 8.0f * (1.0f / 64.0f);                 /* :28-31 */
    else        v = (1.0f + (float)m / 8.0f) * ldexpf(1.0f, (int)e - 7); /* :32-37 */
    return sign ? -v : v;
}

void fp8o_decode_lut(float *out256)
{
    for (int b = 0; b < 256; ++b) out256[b] = decode1((uint8_t)b);
}

/* fp8_matmul.metal:44-92 - float -> byte, on the fp32 bit pattern (exact
 * arithmetic, i.e. the behaviour of the Python twin test_fp8_correctness.py:53-106).
 * NaN input is outside the reference's domain; mapped to 0x7F. */
static uint8_t encode1(float x)
{
    uint32_t bits; memcpy(&bits, &x, 4);
    uint32_t a = bits & 0x7FFFFFFFu;
    uint32_t sign = ((bits >> 31) && a != 0) ? 0x80u : 0u;   /* val < 0 (:46); -0.0 -> 0 */
    if (a > 0x7F800000u) return 0x7F;
    if (a >= 0x43E00000u) return (uint8_t)(sign | 0x7E);     /* >= 448 saturates (:53-55) */
    if (a < 0x3B000000u)  return (uint8_t)sign;              /* < 2^-9 flushes (:58-60) */
    uint32_t e = a >> 23, man = a & 0x7FFFFFu;
    if (a < 0x3C800000u) {                                   /* subnormal (:64-70) */
        uint32_t full = man | 0x800000u, s = 141u - e;       /* val*512 = full * 2^(e-141) */
        uint32_t q = full >> s, r = full & ((1u << s) - 1u), h = 1u << (s - 1u);
        q += (r > h) || (r == h && (q & 1u));
        if (q > 7u) q = 7u;
        return (uint8_t)(sign | q);
    }
    uint32_t q = man >> 20, r = man & 0xFFFFFu;              /* normal (:73-89) */
    q += (r > 0x80000u) || (r == 0x80000u && (q & 1u));
    if (q > 7u) q = 7u;                                      /* clamp, no carry (:81) */
    uint32_t eb = e - 120u;                                  /* e - 127 + 7, in [1,15] */
    if (eb == 15u && q == 7u) q = 6u;                        /* (:87-89) */
    return (uint8_t)(sign | (eb << 3) | q);
}

void fp8o_encode(const float *in, uint8_t *out, size_t n)
{
#pragma omp parallel for schedule(static)
    for (ptrdiff_t i = 0; i < (ptrdiff_t)n; ++i) out[i] = encode1(in[i]);
}

/* fp8_matmul.metal:215-223 - byte -> half; output as IEEE binary16 bit patterns */
static uint16_t half_bits_of_fp8(uint8_t b)
{
    if ((b & 0x7F) == 0x7F) return 0;
    uint16_t sign = (uint16_t)(b & 0x80) << 8;
    unsigned e = (b >> 3) & 0xFu, m = b & 7u;
    if (e == 0) {
        if (m == 0) return sign;
        /* m/8 * 2^-6: normalise into binary16 (min normal 2^-14) */
        int sh = 0; unsigned mm = m;
        while (!(mm & 8u)) { mm <<= 1; ++sh; }
        return (uint16_t)(sign | ((unsigned)(15 - 6 - sh) << 10) | ((mm & 7u) << 7));
    }
    return (uint16_t)(sign | ((e + 8u) << 10) | (m << 7));
}

void fp8o_dequant_f16_bits(const uint8_t *in, uint16_t *out, size_t n)
{
#pragma omp parallel for schedule(static)
    for (ptrdiff_t i = 0; i < (ptrdiff_t)n; ++i) out[i] = half_bits_of_fp8(in[i]);
}

/* fp8_matmul.metal:99-147 (and :155-210 for M == 1).
 * A (M,K) row-major bytes, B (N,K) row-major bytes, C (M,N) float.
 * Same arithmetic as the shader: float accumulate, K unrolled by four with
 * the four products summed before joining the accumulator (:119-141), tail
 * loop (:138-141), then sum * sa * sb (:144-146).  na / nb = number of scale
 * elements (1 = per-tensor, else per-row). */
void fp8o_scaled_mm(const uint8_t *A, const uint8_t *B, float *C,
                    const float *sa, const float *sb,
                    size_t M, size_t N, size_t K, size_t na, size_t nb)
{
    float lut[256];
    fp8o_decode_lut(lut);
    size_t K4 = (K / 4) * 4;
#pragma omp parallel for collapse(2) schedule(static)
    for (ptrdiff_t m = 0; m < (ptrdiff_t)M; ++m) {
        for (ptrdiff_t n = 0; n < (ptrdiff_t)N; ++n) {
            const uint8_t *a = A + (size_t)m * K, *b = B + (size_t)n * K;
            float sum = 0.0f;
            size_t k = 0;
            for (; k < K4; k += 4)
                sum += lut[a[k]] * lut[b[k]] + lut[a[k + 1]] * lut[b[k + 1]]
                     + lut[a[k + 2]] * lut[b[k + 2]] + lut[a[k + 3]] * lut[b[k + 3]];
            for (; k < K; ++k) sum += lut[a[k]] * lut[b[k]];
            float va = (na == 1) ? sa[0] : sa[m];
            float vb = (nb == 1) ? sb[0] : sb[n];
            C[(size_t)m * N + n] = sum * va * vb;
        }
    }
}

/* Same product with double accumulation: the exact dot product (every
 * e4m3 x e4m3 product is exact), used as the yardstick for float error. */
void fp8o_scaled_mm_f64(const uint8_t *A, const uint8_t *B, double *C,
                        const float *sa, const float *sb,
                        size_t M, size_t N, size_t K, size_t na, size_t nb)
{
    float lut[256];
    fp8o_decode_lut(lut);
#pragma omp parallel for collapse(2) schedule(static)
    for (ptrdiff_t m = 0; m < (ptrdiff_t)M; ++m) {
        for (ptrdiff_t n = 0; n < (ptrdiff_t)N; ++n) {
            const uint8_t *a = A + (size_t)m * K, *b = B + (size_t)n * K;
            double sum = 0.0;
            for (size_t k = 0; k < K; ++k) sum += (double)lut[a[k]] * (double)lut[b[k]];
            double va = (na == 1) ? sa[0] : sa[m];
            double vb = (nb == 1) ? sb[0] : sb[n];
            C[(size_t)m * N + n] = sum * va * vb;
        }
    }
}

int fp8o_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
