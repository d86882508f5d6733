// How long does one `buffer_load_dwordx4 ... lds` (1 KiB LDS-DMA) hold the issuing wave, alone and between MFMAs, with 1 / 2 / 4 waves of the CU issuing?
// One workgroup of 4 waves per CU (one wave per SIMD: 512-register launch bounds via a dummy asm clobber), source rows L2-resident.
//   hipcc --offload-arch=gfx950 -O3 dma_probe.hip -o dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP16(x) x x x x x x x x x x x x x x x x
#define DMA "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tbuffer_load_dwordx4 %[voff], %[rs], 0 offen lds\n\tv_add_u32 %[voff], 0x2000, %[voff]\n\t"
#define MFMA(a) "v_mfma_scale_f32_16x16x128_f8f6f4 a[" #a "], v[136:143], v[144:151], a[" #a "], v120, v120 op_sel_hi:[0,0,0]\n\t"
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(256) void probe(unsigned long long *out, const unsigned char *src, int nwaves)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[128 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long base = (unsigned long long)(src + (size_t)(blockIdx.x & 31) * 65536 * 4);
    u32x4 rs = {(unsigned)base, (unsigned)(base >> 32) & 0xFFFFu, 0x7FFFFFFFu, 0x00020000u};
    rs[0] = __builtin_amdgcn_readfirstlane(rs[0]); rs[1] = __builtin_amdgcn_readfirstlane(rs[1]);
    unsigned voff = wave * 65536 + lane * 16;
    unsigned m0v = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char *)lds + wave * 32768;
    unsigned long long c0 = 0, c1 = 0;
    asm volatile("v_mov_b32 v120, 0x7f7f7f7f" ::: "v120");
    if (wave < nwaves) {
#define RUN(BODY) asm volatile("s_mov_b32 m0, %[m0v]\n\ts_barrier\n\ts_memtime %[c0]\n\ts_waitcnt lgkmcnt(0)\n\t" BODY "s_memtime %[c1]\n\ts_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(0)\n\t" \
        : [c0] "=&s"(c0), [c1] "=&s"(c1), [voff] "+v"(voff) : [rs] "s"(rs), [m0v] "s"(m0v) \
        : "memory", "scc", "m0", "v120", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", \
          "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "v255", "a255")
        if (V == 0) RUN(REP16(DMA));
        if (V == 1) RUN(REP16(MFMA(0:3) MFMA(4:7) MFMA(8:11)));
        if (V == 2) RUN(REP16(MFMA(0:3) DMA MFMA(4:7) MFMA(8:11)));
        if (V == 3) RUN(REP16(MFMA(0:3) DMA MFMA(4:7) DMA MFMA(8:11) DMA));
    } else {
        asm volatile("s_barrier" ::: "memory");
    }
    if (lane == 0 && wave < nwaves) out[blockIdx.x * 4 + wave] = c1 - c0;
}

int main()
{
    const int blocks = 256;
    unsigned long long *d; unsigned char *src;
    hipMalloc(&d, blocks * 4 * 8); hipMalloc(&src, 32u * 65536 * 4 + (1u << 20)); hipMemset(src, 1, 32u * 65536 * 4 + (1u << 20));
    std::vector<unsigned long long> h(blocks * 4);
    const char *names[] = {"16 x DMA back to back", "48 MFMA (no DMA)", "16 x [MFMA DMA MFMA MFMA]", "16 x [MFMA DMA] x 3"};
    for (int v = 0; v < 4; ++v)
        for (int nw = 1; nw <= 4; nw *= 2) {
            for (int r = 0; r < 3; ++r) {
                hipMemset(d, 0, blocks * 4 * 8);
                switch (v) {
                    case 0: hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, d, src, nw); break;
                    case 1: hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, d, src, nw); break;
                    case 2: hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, d, src, nw); break;
                    case 3: hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(256), 0, 0, d, src, nw); break;
                }
                hipDeviceSynchronize();
            }
            hipMemcpy(h.data(), d, blocks * 4 * 8, hipMemcpyDeviceToHost);
            double s = 0; int n = 0;
            for (auto x : h) if (x) { s += x; ++n; }
            const int ndma = v == 0 ? 16 : (v == 2 ? 16 : (v == 3 ? 48 : 0));
            printf("%-28s %d wave(s) issuing: %8.1f cycles total", names[v], nw, s / n);
            if (v == 0) printf("  = %6.1f per DMA", s / n / 16);
            if (v >= 2) printf("  = %6.1f over the 48 MFMAs' 1536+ -> %6.1f per DMA", s / n, 0.0);
            printf("\n");
        }
    return 0;
}
