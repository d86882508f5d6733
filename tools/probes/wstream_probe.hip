// How fast can 256 CUs stream a cold weight matrix (N = 4096 rows x K = 14336 bytes, config C2 / "decode") as a function of
//   (a) the instruction: `buffer_load_dwordx4 ... lds` (LDS-DMA, what the tile kernels use) or a plain non-temporal load into VGPRs (the vec-mat), and
//   (b) how many CONTIGUOUS bytes of one row an instruction (and a K-step) takes: 128 B (the tile kernels' BK = 128: an instruction = 8 rows x one line),
//       512 B (2 rows x 4 lines) or 1 KiB (1 row x 8 lines)?
// Every workgroup (4 waves) owns 32 rows x 7168 bytes (128 row tiles x 2 K slices = 256 workgroups = one per CU: 64 KiB of LDS + launch bounds keep it at one),
// DEPTH instructions issued back to back per wave, then drained (56 = the wave's whole share at once); six matrices in rotation (352 MB: cold).
// Prints in-kernel bytes / clock / CU (median CU) and the launch-to-end rate from events (which contains the ~4 us dispatch floor).
//   hipcc --offload-arch=gfx950 -O3 wstream_probe.hip -o wstream_probe && ./wstream_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kN = 4096, kK = 14336, kRows = 32, kSlice = 7168;

// RUN = contiguous bytes of a row per instruction (128 / 512 / 1024); DMA = 1: global -> LDS, 0: global -> VGPR (nt)
template <int RUN, int DMA, int DEPTH>
__global__ __launch_bounds__(256) void wstream(const uint8_t *W, unsigned long long *cyc, uint32_t *out)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[64 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile = blockIdx.x >> 1, slice = blockIdx.x & 1;
    const uint32_t base = (uint32_t)tile * kRows * kK + (uint32_t)slice * kSlice;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)W, 0, kN * kK, 0x00020000);
    constexpr int kRowsPerInstr = 1024 / RUN;              // 8, 2, 1
    constexpr int kLanesPerRow = RUN / 16;                 // 8, 32, 64
    constexpr int kInstrPerStep = kRows / kRowsPerInstr;   // instructions that cover all 32 rows at one K position: 4, 16, 32
    constexpr int kSteps = kSlice / RUN;                   // 56, 14, 7
    constexpr int kTotal = kInstrPerStep * kSteps / 4;     // per wave: 56 in every form
    const uint32_t lrow = (uint32_t)(lane / kLanesPerRow), lcol = (uint32_t)(lane % kLanesPerRow) * 16u;
    u32x4 acc = {0u, 0u, 0u, 0u};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < kTotal; i += DEPTH) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const int j = (i + u) * 4 + wave;              // instruction index of the workgroup: step-major, then row group
            const int step = j / kInstrPerStep, grp = j % kInstrPerStep;
            const uint32_t off = base + ((uint32_t)grp * kRowsPerInstr + lrow) * kK + (uint32_t)step * RUN + lcol;
            if (i + u < kTotal) {
                if (DMA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(lds + wave * 16384 + ((i + u) & 15) * 1024), 16, off, 0, 0, 0);
                else v[u] = __builtin_nontemporal_load((const u32x4 *)(W + off));
            }
        }
        if (!DMA) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u)
                if (i + u < kTotal) acc ^= v[u];
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[threadIdx.x] = acc[0];
}

template <int RUN, int DMA, int DEPTH>
static void run(uint8_t *const *Ws, int nW, unsigned long long *d_cyc, uint32_t *d_out)
{
    std::vector<unsigned long long> h(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f; double med = 0;
    for (int r = 0; r < 13; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((wstream<RUN, DMA, DEPTH>), dim3(256), dim3(256), 0, 0, Ws[r % nW], d_cyc, d_out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r == 0) continue;
        if (ms < best) {
            best = ms;
            hipMemcpy(h.data(), d_cyc, 256 * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            med = (double)h[128];
        }
    }
    const double bytes_wg = (double)kRows * kSlice;
    printf("%-8s %4d contiguous bytes per row and instruction, %2d in flight per wave: %6.2f B/clk/CU in-kernel (median CU, %6.0f clk); launch to end %6.2f us = %5.2f TB/s\n",
           DMA ? "LDS-DMA" : "VGPR nt", RUN, DEPTH, bytes_wg / med, med, best * 1e3, bytes_wg * 256 / (best * 1e-3) / 1e12);
}

int main()
{
    const int nW = 6;
    uint8_t *Ws[nW]; unsigned long long *d_cyc; uint32_t *d_out;
    for (int i = 0; i < nW; ++i) { if (hipMalloc(&Ws[i], (size_t)kN * kK) != hipSuccess) { printf("alloc failed\n"); return 1; } hipMemset(Ws[i], i + 1, (size_t)kN * kK); }
    hipMalloc(&d_cyc, 256 * 8); hipMalloc(&d_out, 4096);
    hipDeviceSynchronize();
    run<128, 1, 8>(Ws, nW, d_cyc, d_out);  run<128, 1, 14>(Ws, nW, d_cyc, d_out);  run<128, 1, 28>(Ws, nW, d_cyc, d_out);  run<128, 1, 56>(Ws, nW, d_cyc, d_out);
    run<512, 1, 8>(Ws, nW, d_cyc, d_out);  run<512, 1, 14>(Ws, nW, d_cyc, d_out);  run<512, 1, 28>(Ws, nW, d_cyc, d_out);  run<512, 1, 56>(Ws, nW, d_cyc, d_out);
    run<1024, 1, 8>(Ws, nW, d_cyc, d_out); run<1024, 1, 14>(Ws, nW, d_cyc, d_out); run<1024, 1, 28>(Ws, nW, d_cyc, d_out); run<1024, 1, 56>(Ws, nW, d_cyc, d_out);
    run<128, 0, 8>(Ws, nW, d_cyc, d_out);  run<128, 0, 14>(Ws, nW, d_cyc, d_out);  run<128, 0, 28>(Ws, nW, d_cyc, d_out);
    run<512, 0, 8>(Ws, nW, d_cyc, d_out);  run<512, 0, 14>(Ws, nW, d_cyc, d_out);  run<512, 0, 28>(Ws, nW, d_cyc, d_out);
    run<1024, 0, 8>(Ws, nW, d_cyc, d_out); run<1024, 0, 14>(Ws, nW, d_cyc, d_out); run<1024, 0, 28>(Ws, nW, d_cyc, d_out);
    return 0;
}
