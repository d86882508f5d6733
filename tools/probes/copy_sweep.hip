// Ceilings of the two cast shapes at 2^30 elements: "quantize" moves 4 bytes in per byte out (5 B/elem), "dequant" 1 byte in per 2 bytes out
// (3 B/elem).  No arithmetic - only the memory instructions the cast kernels issue, in their present form and in the forms a re-layout could reach:
//   Q1  16-B load per lane, 4-B store per lane (fp8mi_cast.hip encode_kernel<F32>)        Q2  four 16-B loads, ONE 16-B store per lane
//   D1  16-B load, two 16-B stores at a 32-byte lane stride (dequant_kernel)               D2  16-B load, two 16-B stores each a contiguous KiB per wave
// swept over the grid (0 = one pass per workgroup) and the store policy.   hipcc --offload-arch=gfx950 -O3 copy_sweep.hip -o copy_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool NTS>
__global__ __launch_bounds__(256) void k_copy(const u32x4 *__restrict__ src, uint32_t *__restrict__ dst, size_t n16 /* 16-byte input pieces */)
{
    const size_t stride = (size_t)gridDim.x * 256;
    auto st4 = [&](uint32_t *p, uint32_t v) { if (NTS) __builtin_nontemporal_store(v, p); else *p = v; };
    auto st16 = [&](u32x4 *p, u32x4 v) { if (NTS) __builtin_nontemporal_store(v, p); else *p = v; };
    if (MODE == 0) {   // Q1
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
            u32x4 v = __builtin_nontemporal_load(src + i);
            st4(dst + i, v[0] ^ v[1] ^ v[2] ^ v[3]);
        }
    } else if (MODE == 1) {   // Q2: pieces i, i + 256, i + 512, i + 768 of a 16-KiB block -> 16 output bytes of the lane
        for (size_t b = blockIdx.x; b * 1024 < n16; b += gridDim.x) {
            const size_t i = b * 1024 + threadIdx.x;
            u32x4 v0 = __builtin_nontemporal_load(src + i), v1 = __builtin_nontemporal_load(src + i + 256), v2 = __builtin_nontemporal_load(src + i + 512),
                  v3 = __builtin_nontemporal_load(src + i + 768);
            st16((u32x4 *)dst + b * 256 + threadIdx.x, u32x4{v0[0] ^ v0[3], v1[1] ^ v1[2], v2[2] ^ v2[0], v3[3] ^ v3[1]});
        }
    } else if (MODE == 2) {   // D1
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
            u32x4 v = __builtin_nontemporal_load(src + i);
            st16((u32x4 *)dst + 2 * i, v);
            st16((u32x4 *)dst + 2 * i + 1, v ^ 1u);
        }
    } else {   // D2: the wave's two stores are each one contiguous KiB
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
            u32x4 v = __builtin_nontemporal_load(src + i);
            const size_t w = i >> 6, l = i & 63;
            st16((u32x4 *)dst + w * 128 + l, v);
            st16((u32x4 *)dst + w * 128 + 64 + l, v ^ 1u);
        }
    }
}

int main()
{
    const size_t n = 1ull << 30;
    void *a, *b;
    hipMalloc(&a, n * 4); hipMalloc(&b, n * 2); hipMemset(a, 1, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[] = {"Q1 quantize now   (16 B in, 4 B out per lane)", "Q2 quantize ideal (4 x 16 B in, 16 B out)", "D1 dequant now    (16 B in, 2 x 16 B at 32-B stride)",
                           "D2 dequant ideal  (16 B in, 2 x 16 B contiguous)"};
    for (int mode = 0; mode < 4; ++mode) {
        const bool q = mode < 2;
        const size_t n16 = q ? n * 4 / 16 : n / 16;              // input pieces
        const double bytes = q ? 5.0 * n : 3.0 * n;
        struct Row { double gbs; int grid; bool nts; };
        std::vector<Row> rows;
        for (int nts = 0; nts < 2; ++nts)
            for (int grid : {0, 1024, 2048, 4096, 8192, 16384, 65536, 262144}) {
                const size_t per_wg = mode == 1 ? 1024 : 256;
                int g = grid ? grid : (int)((n16 + per_wg - 1) / per_wg);
                auto run = [&]() {
                    switch (mode * 2 + nts) {
                        case 0: hipLaunchKernelGGL((k_copy<0, false>), dim3(g), dim3(256), 0, 0, (const u32x4 *)a, (uint32_t *)b, n16); break;
                        case 1: hipLaunchKernelGGL((k_copy<0, true>), dim3(g), dim3(256), 0, 0, (const u32x4 *)a, (uint32_t *)b, n16); break;
                        case 2: hipLaunchKernelGGL((k_copy<1, false>), dim3(g), dim3(256), 0, 0, (const u32x4 *)a, (uint32_t *)b, n16); break;
                        case 3: hipLaunchKernelGGL((k_copy<1, true>), dim3(g), dim3(256), 0, 0, (const u32x4 *)a, (uint32_t *)b, n16); break;
                        case 4: hipLaunchKernelGGL((k_copy<2, false>), dim3(g), dim3(256), 0, 0, (const u32x4 *)a, (uint32_t *)b, n16); break;
                        case 5: hipLaunchKernelGGL((k_copy<2, true>), dim3(g), dim3(256), 0, 0, (const u32x4 *)a, (uint32_t *)b, n16); break;
                        case 6: hipLaunchKernelGGL((k_copy<3, false>), dim3(g), dim3(256), 0, 0, (const u32x4 *)a, (uint32_t *)b, n16); break;
                        case 7: hipLaunchKernelGGL((k_copy<3, true>), dim3(g), dim3(256), 0, 0, (const u32x4 *)a, (uint32_t *)b, n16); break;
                    }
                };
                run(); run();
                hipEventRecord(e0);
                for (int r = 0; r < 6; ++r) run();
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                rows.push_back({bytes / (ms * 1e-3 / 6) / 1e9, g, (bool)nts});
            }
        std::sort(rows.begin(), rows.end(), [](const Row &x, const Row &y) { return x.gbs > y.gbs; });
        printf("== %s ==\n", names[mode]);
        for (auto &r : rows) printf("   %7.1f GB/s  (%7.1f us)  grid %8d  stores %s\n", r.gbs, bytes / r.gbs / 1e3, r.grid, r.nts ? "nt" : "default");
    }
    return 0;
}
