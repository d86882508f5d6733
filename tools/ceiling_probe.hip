// Measured ceilings for bench.py (NOT part of the product library): what this MI355X sustains on
//   (1) fp8 MFMA issued back to back from registers (v_mfma_scale_f32_16x16x128_f8f6f4, 16 independent accumulators
//       per wave, 2 workgroups per CU), on operand bytes the caller provides (bench.py passes weight-like bytes);
//   (2) a streaming read of a buffer (16 B per lane, non-temporal; grid and loads in flight per lane as swept in round 3).
// Built as tools/libceiling_probe.so by __graft_entry__.build() / bench.py; C ABI, launches only (the caller times them
// with events on the stream it passes).
//   hipcc --offload-arch=gfx950 -O3 -fPIC -shared tools/ceiling_probe.hip -o tools/libceiling_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k_mfma(const i32x8 *in, f32x4 *out, int iters)
{
    const int l = threadIdx.x;
    i32x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(l + 64 * i) % 1024]; b[i] = in[(l + 64 * (i + 4)) % 1024]; }
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i], b[j], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    f32x4 s = {0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
    out[(size_t)blockIdx.x * blockDim.x + l] = s;
}

// Grid-strided streaming read, U 16-byte loads in flight per lane.  Which (grid, U) is fastest was swept in round 3 (tools/probes/read_sweep.hip,
// profiles/r03_read_sweep.txt: 160 configurations per buffer size): a 56 MiB buffer read once per launch is fastest with 2048 workgroups and ONE
// load in flight per lane (6.1-6.2 TB/s; round 2's point - 4 in flight - read 5.4), a 2 GiB buffer with 256 workgroups and 8 in flight (7.1-7.3 TB/s).
template <int U>
__global__ __launch_bounds__(256) void k_read(const u32x4 *src, uint32_t *sink, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride * U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u] = u32x4{0, 0, 0, 0};
            if (i + stride * u < n16) v[u] = __builtin_nontemporal_load(src + i + stride * u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

extern "C" {

// operands: 32 KiB of fp8 bytes (device); out: blocks * waves * 64 * 16 bytes (device).  flops = probe_mfma_flops(...)
int probe_mfma(const void *operands, void *out, int blocks, int waves, int iters, void *stream)
{
    hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(waves * 64), 0, (hipStream_t)stream, (const i32x8 *)operands, (f32x4 *)out, iters);
    return (int)hipGetLastError();
}

double probe_mfma_flops(int blocks, int waves, int iters) { return (double)blocks * waves * iters * 16 * 2.0 * 16 * 16 * 128; }

// in_flight: 1, 4 or 8 loads per lane
int probe_read(const void *src, void *sink, size_t bytes, int grid, int in_flight, void *stream)
{
    if (in_flight == 1) hipLaunchKernelGGL(k_read<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const u32x4 *)src, (uint32_t *)sink, bytes / 16);
    else if (in_flight == 8) hipLaunchKernelGGL(k_read<8>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const u32x4 *)src, (uint32_t *)sink, bytes / 16);
    else hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const u32x4 *)src, (uint32_t *)sink, bytes / 16);
    return (int)hipGetLastError();
}

// revision of this file's C signatures (bench.py PROBE_ABI): 2 = probe_read(buf, sink, bytes, grid, in_flight, stream)
int probe_abi_version() { return 2; }
}  // extern "C"
