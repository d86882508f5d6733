// Probe: per-CU cost of moving L2-resident bytes on-chip, LDS-DMA vs register loads.
//   hipcc --offload-arch=gfx950 -O3 tools/load_probe.hip -o tools/load_probe && tools/load_probe
// One workgroup per CU (256 blocks), W waves each, every wave issues R x 1 KiB
// loads per round from a small (L2-resident) buffer, ROUNDS rounds.  Reports
// cycles per wave-instruction and GB/s per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

// PAT (MODE 0 only): 0 = 1 KiB contiguous per instruction; 1 = 8 rows x 128 B, rows 4 KiB apart (the GEMM's staging
// shape); 2 = the same with the GEMM's XOR swizzle of the 16-byte chunks inside each line; 3 = 4 rows x 256 B
template <int MODE, int R, int PAT = 0>  // 0: buffer_load ... lds, 1: global_load_dwordx4 -> VGPR (+xor sink), 2: VGPR + ds_write_b128
__global__ __launch_bounds__(1024) void k(const uint8_t *src, uint32_t *sink, unsigned long long *cyc, int rounds, int span)
{
    __shared__ __attribute__((aligned(16))) uint8_t smem[64 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, span, 0x00020000);
    uint32_t acc = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    uint32_t off = (uint32_t)((blockIdx.x * 7919u + wave * 131u) * 1024u);
    for (int r = 0; r < rounds; ++r) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                uint32_t o = (off + j * 1024u * nw + lane * 16u) % (uint32_t)span;
                if (PAT == 1 || PAT == 2) {
                    const uint32_t row = lane >> 3, ch = (lane & 7) ^ (PAT == 2 ? ((wave & 1) * 4 + (lane >> 4)) & 7 : 0);
                    const uint32_t q = (uint32_t)(r * R + j) * nw + wave;  // the workgroup walks 8-row x 4-KiB blocks
                    o = (blockIdx.x * (1u << 20) + (q / 32u) * 32768u + row * 4096u + (q % 32u) * 128u + ch * 16u) % (uint32_t)span;
                } else if (PAT == 3) {
                    const uint32_t row = lane >> 4, ch = lane & 15;
                    o = ((off / 1024u + j * nw) % 16u * 256u + ((off / 16384u) % 16u) * 16384u * 8u + row * 4096u + ch * 16u) % (uint32_t)span;
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(smem + ((wave * R + j) % 64) * 1024), 16, (int)o, 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            u32x4 v[R];
#pragma unroll
            for (int j = 0; j < R; ++j) {
                uint32_t o = (off + j * 1024u * nw + lane * 16u) % (uint32_t)span;
                v[j] = *(const u32x4 *)(src + o);
            }
#pragma unroll
            for (int j = 0; j < R; ++j) {
                if (MODE == 1) acc ^= v[j][0] ^ v[j][1] ^ v[j][2] ^ v[j][3];
                else *(u32x4 *)(smem + ((wave * R + j) % 64) * 1024 + lane * 16) = v[j];
            }
        }
        off += 1024u * nw * R;
    }
    if (MODE == 2) { __syncthreads(); acc = *(uint32_t *)(smem + threadIdx.x * 4); }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int R, int PAT = 0>
void run(const char *name, int waves, const uint8_t *src, uint32_t *sink, unsigned long long *cyc, int span, int grid = 256)
{
    const int rounds = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, R, PAT>), grid, waves * 64, 0, 0, src, sink, cyc, PAT ? rounds : 20, span);  // warm-up (same footprint for the row patterns)
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, R, PAT>), grid, waves * 64, 0, 0, src, sink, cyc, rounds, span);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += c; avg /= grid;
    double instr_per_wave = (double)rounds * R;
    double bytes_cu = instr_per_wave * waves * 1024.0;
    printf("%-26s waves/CU %2d  R %2d: %7.1f cyc per wave-instr, %6.1f GB/s per CU (%5.2f TB/s chip), kernel %.1f us\n", name, waves, R,
           avg / instr_per_wave, bytes_cu / (ms * 1e-3) / 1e9, bytes_cu * grid / (ms * 1e-3) / 1e12, ms * 1e3);
}

int main(int argc, char **)
{
    const int span = 2 << 20;  // 2 MiB: L2-resident per XCD
    uint8_t *src; uint32_t *sink; unsigned long long *cyc;
    hipMalloc(&src, span + 65536); hipMemset(src, 1, span + 65536); hipMalloc(&sink, 4); hipMalloc(&cyc, 256 * 8);
    for (int w : {1, 2, 4, 8, 16}) {
        run<0, 6>("LDS-DMA (buffer_load lds)", w, src, sink, cyc, span);
        run<1, 6>("global_load_dwordx4->VGPR", w, src, sink, cyc, span);
        run<2, 6>("VGPR + ds_write_b128", w, src, sink, cyc, span);
    }
    for (int w : {4, 8}) {
        run<0, 16>("LDS-DMA (buffer_load lds)", w, src, sink, cyc, span);
        run<1, 16>("global_load_dwordx4->VGPR", w, src, sink, cyc, span);
        run<2, 16>("VGPR + ds_write_b128", w, src, sink, cyc, span);
    }
    for (int w : {4, 8}) {
        run<0, 16, 0>("DMA 1 KiB contiguous", w, src, sink, cyc, span);
        run<0, 16, 1>("DMA 8 rows x 128 B", w, src, sink, cyc, span);
        run<0, 16, 2>("DMA 8 rows x 128 B swizzled", w, src, sink, cyc, span);
        run<0, 16, 3>("DMA 4 rows x 256 B", w, src, sink, cyc, span);
    }
    // how much of the per-CU rate is bytes-in-flight / latency: the same stream from L2 (2 MiB), from the Infinity
    // Cache (96 MiB span, warmed by the first launch) and with 64 instead of 256 CUs active
    if (argc > 1) {
        uint8_t *big; hipMalloc(&big, (size_t)1 << 30);
        hipMemset(big, 1, (size_t)1 << 30);
        for (int sp : {2 << 20, 16 << 20, 96 << 20, 1 << 30})
            for (int grid : {64, 256}) {
                printf("span %4d MiB grid %3d: ", sp >> 20, grid);
                run<0, 16, 2>("DMA 8x128 swz", 8, big, sink, cyc, sp, grid);
            }
    }
    return 0;
}
