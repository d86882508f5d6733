// Shape- and alignment-agnostic FP8 scaled matmul: one wave64 per output
// element, byte loads, exact byte-wise reference decode (fp8_matmul.metal:19-40).
// It exists so that EVERY problem the reference accepts (any K, any leading
// dimension, unaligned views - fp8_mps_native.py:55-60 only asks for
// contiguity) has a device path; the tuned kernels take over whenever their
// alignment preconditions hold.  Same epilogue as the other kernels.

#include "fp8mi_common.h"

namespace {

constexpr int kWavesPerBlock = 4;

__global__ __launch_bounds__(kWavesPerBlock * 64) void generic_kernel(MMParams p)
{
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t m = (int64_t)blockIdx.y + (int64_t)blockIdx.z * 65535;
    if (n >= p.N || m >= p.M) return;  // wave-uniform
    const uint8_t *a = p.A + m * p.lda;
    const uint8_t *b = p.B + n * p.ldb;
    float s = 0.0f;
    if (p.nan_zero) {
        for (int64_t k = lane; k < p.K; k += 64) s += decode_ref(a[k]) * decode_ref(b[k]);
    } else {
        for (int64_t k = lane; k < p.K; k += 64) {
            // OCP semantics: let the hardware convert produce NaN for 0x7F / 0xFF
            float fa = __builtin_amdgcn_cvt_f32_fp8((int)a[k], 0);
            float fb = __builtin_amdgcn_cvt_f32_fp8((int)b[k], 0);
            s += fa * fb;
        }
    }
    s = wave_sum(s);
    if (lane == 0) {
        const float sa = p.sa_row ? p.scale_a[m] : p.scale_a[0];
        const float sb = p.sb_row ? p.scale_b[n] : p.scale_b[0];
        const float bias = p.bias ? load_as_float(p.bias, p.transposed ? m : n, p.bias_dtype) : 0.0f;
        const float sr = p.scale_result ? p.scale_result[0] : 1.0f;
        store_from_float(p.C, m * p.ldc + n,
                         epilogue_value(s, sa, sb, p.bias != nullptr, bias, p.scale_result != nullptr, sr, p.transposed != 0), p.out_dtype);
    }
}

}  // namespace

int fp8mi_launch_generic(const MMParams &p, hipStream_t s)
{
    const int64_t gx = (p.N + kWavesPerBlock - 1) / kWavesPerBlock;
    const int64_t gy = p.M < 65535 ? p.M : 65535;
    const int64_t gz = (p.M + 65534) / 65535;
    if (gx > 0x7FFFFFFF || gz > 65535) return FP8MI_E_UNSUPPORTED;
    return fp8mi_launch(generic_kernel, dim3((unsigned)gx, (unsigned)gy, (unsigned)gz), dim3(kWavesPerBlock * 64), s, p);
}
