// Streaming-read ceiling of one MI355X, swept: how fast can ANY kernel read a 56 MiB buffer once per launch (six rotating buffers: the size of
// config C2's weight matrix, > 256 MiB in rotation) and a 2 GiB buffer?  Sweeps workgroups per launch, 16-byte loads in flight per lane, cache policy
// (nt / default) and the walk (grid-strided 4 KiB pieces / one contiguous span per workgroup).  Round 2's bench probe was ONE point of this table
// (grid 2048, 4 in flight, nt, strided); the judge asked whether the point was the ceiling.   hipcc --offload-arch=gfx950 -O3 read_sweep.hip -o read_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT, bool SPAN>
__global__ __launch_bounds__(256) void k_read(const u32x4 *__restrict__ src, uint32_t *sink, size_t n16)
{
    u32x4 acc = {0, 0, 0, 0};
    if (SPAN) {   // workgroup b reads the contiguous span [b, b+1) * n16 / grid, U x 4 KiB per trip
        const size_t per = (n16 + gridDim.x - 1) / gridDim.x, lo = (size_t)blockIdx.x * per, hi = lo + per < n16 ? lo + per : n16;
        for (size_t i = lo + threadIdx.x; i < hi; i += (size_t)256 * U) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t j = i + (size_t)256 * u;
                v[u] = u32x4{0, 0, 0, 0};
                if (j < hi) v[u] = NT ? __builtin_nontemporal_load(src + j) : src[j];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u];
        }
    } else {      // grid-strided: piece p of 4 KiB goes to workgroup p % grid
        const size_t stride = (size_t)gridDim.x * 256;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride * U) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t j = i + stride * u;
                v[u] = u32x4{0, 0, 0, 0};
                if (j < n16) v[u] = NT ? __builtin_nontemporal_load(src + j) : src[j];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc ^= v[u];
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

typedef void (*kern_t)(const u32x4 *, uint32_t *, size_t);
template <int U> kern_t pick(bool nt, bool span)
{
    if (nt) return span ? k_read<U, true, true> : k_read<U, true, false>;
    return span ? k_read<U, false, true> : k_read<U, false, false>;
}

int main(int argc, char **argv)
{
    // argv[2] = size of the rotating buffers in MiB (default 56 = config C2's weights; 16 = config C1's), as many of them as exceed 320 MiB
    const size_t small = (size_t)(argc > 2 ? atoi(argv[2]) : 56) << 20, big = 2ull << 30;
    const int nbuf = (int)((320u << 20) / small) + 1;
    std::vector<void *> bufs(nbuf);
    for (auto &b : bufs) { hipMalloc(&b, small); hipMemset(b, 1, small); }
    void *bigbuf; hipMalloc(&bigbuf, big); hipMemset(bigbuf, 1, big);
    uint32_t *sink; hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Row { double gbs; int grid, U; bool nt, span; };
    for (int which = 0; which < (argc > 2 ? 1 : 2); ++which) {
        std::vector<Row> rows;
        const size_t bytes = which ? big : small;
        const int reps = which ? 12 : 360;
        for (int U : {1, 2, 4, 8, 16})
            for (int nt = 0; nt < 2; ++nt)
                for (int span = 0; span < 2; ++span)
                    for (int grid : {256, 512, 1024, 2048, 4096, 8192, 16384, 0}) {
                        int g = grid ? grid : (int)(bytes / ((size_t)U * 4096));   // 0 = one trip per workgroup
                        if (g > (1 << 20)) continue;
                        kern_t k = U == 1 ? pick<1>(nt, span) : U == 2 ? pick<2>(nt, span) : U == 4 ? pick<4>(nt, span) : U == 8 ? pick<8>(nt, span) : pick<16>(nt, span);
                        auto run = [&](int i) { hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, (const u32x4 *)(which ? bigbuf : bufs[i % nbuf]), sink, bytes / 16); };
                        for (int i = 0; i < 12; ++i) run(i);
                        hipEventRecord(e0);
                        for (int i = 0; i < reps; ++i) run(i);
                        hipEventRecord(e1); hipEventSynchronize(e1);
                        float ms; hipEventElapsedTime(&ms, e0, e1);
                        rows.push_back({bytes / (ms * 1e-3 / reps) / 1e9, g, U, (bool)nt, (bool)span});
                    }
        std::sort(rows.begin(), rows.end(), [](const Row &a, const Row &b) { return a.gbs > b.gbs; });
        printf("== %s, back-to-back launches (events around %d launches: includes the launch boundary, as bench.py's step does) ==\n",
               which ? "2 GiB buffer" : "rotating buffers (> 320 MiB in all), one read once per launch; MiB each =", reps);
        if (!which) printf("   buffer size %zu MiB x %d buffers\n", small >> 20, nbuf);
        for (size_t i = 0; i < rows.size(); ++i)
            if (argc > 1 || i < 12 || i + 4 >= rows.size() || (rows[i].grid == 2048 && rows[i].U == 4 && rows[i].nt && !rows[i].span))
                printf("  #%3zu  %7.1f GB/s  (%6.2f us per launch)  grid %6d  in flight %2d  %s  %s%s\n", i + 1, rows[i].gbs, bytes / rows[i].gbs / 1e3,
                       rows[i].grid, rows[i].U, rows[i].nt ? "nt     " : "default", rows[i].span ? "contiguous span" : "grid-strided   ",
                       (rows[i].grid == 2048 && rows[i].U == 4 && rows[i].nt && !rows[i].span) ? "   <- round 2's bench probe" : "");
    }
    return 0;
}
