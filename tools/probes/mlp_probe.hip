// Memory-level parallelism of ONE compute unit's global -> LDS path (gfx950): how many bytes per clock a CU takes in as a function of
//   (a) where the lines come from - the XCD's L2 (lines every CU of the XCD re-reads), the Infinity Cache (each CU its own 512 KiB of a 128 MiB
//       set that was just streamed), HBM (each CU its own 4 MiB of a cold 1 GiB span) - and
//   (b) how many 128-byte lines it keeps in flight: 4 waves x DEPTH `buffer_load_dwordx4 ... lds` instructions (1 KiB = 8 lines each),
// with all 256 CUs pulling at once (one 256-thread workgroup per CU: 64 KiB of LDS + launch bounds keep it at one).
// If the rate stops growing with DEPTH, the CU's L1 has a cap on outstanding line requests, and rate = cap x 128 B / latency: the tile kernels'
// K loops (mean L2 read latency 400-500 cycles at ~0.8 hit rate, profiles/r03_pmc_gemm_*.txt) are then bound by that product, not by issue.
// MIX 1: every 4th instruction of a wave reads its cold HBM span, the other three the XCD-shared L2 span (a tile kernel's ~0.8 hit rate).
// MIX 2: the same stream, but the cold lines were touched PFD x 32 iterations earlier by a one-dword-per-line prefetch load of the same wave (the tile
//        kernels' L2 prefetch): the demand stream then finds them in the L2, the misses ride on loads whose data nobody waits for.
// W: waves per workgroup (4 = one per SIMD, 8 = two).
//   hipcc --offload-arch=gfx950 -O3 mlp_probe.hip -o mlp_probe && ./mlp_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;

template <int DEPTH, int MIX, int W = 4, int PFD = 2>
__global__ __launch_bounds__(W * 64) void mlp(const uint8_t *hot, const uint8_t *src, size_t wg_stride, uint32_t span, int iters, unsigned long long *cyc)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[64 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(src + (size_t)blockIdx.x * wg_stride), 0, (int)span, 0x00020000);
    __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void *)hot, 0, 256 * 1024, 0x00020000);
    const uint32_t q = span / W, voff = (uint32_t)lane * 16u;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    const unsigned long long pb = (unsigned long long)(src + (size_t)blockIdx.x * wg_stride);
    u32x4 rpf = {(uint32_t)pb, (uint32_t)(pb >> 32) & 0xFFFFu, span, 0x00020000u};
    rpf[0] = __builtin_amdgcn_readfirstlane(rpf[0]); rpf[1] = __builtin_amdgcn_readfirstlane(rpf[1]); rpf[2] = __builtin_amdgcn_readfirstlane(rpf[2]);
    uint32_t sink = 0, ssink = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        lds_void *dst = (lds_void *)(lds + wave * (65536 / W) + (it & (64 / W - 1)) * 1024);
        const uint32_t soff = (uint32_t)wave * q + (((uint32_t)it * 1024u) & (q - 1));
        if (MIX == 2 && (it & 31) == 0) {   // the 8 cold pieces of the block PFD blocks ahead: lane -> (piece lane >> 3, line lane & 7)
            const uint32_t j = (uint32_t)it + 32u * PFD + 3u + 4u * (uint32_t)(lane >> 3);
            const uint32_t off = (uint32_t)wave * q + ((j * 1024u) & (q - 1)) + (uint32_t)(lane & 7) * 128u;
            asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "+v"(sink) : "v"(off), "s"(rpf) : "memory");
        }
        if (MIX == 3 && (it & 3) == 0) {   // the same lines, touched through the SCALAR cache path instead (s_load: K$ -> L2, not the CU's vector L1):
            // per 4 iterations one cold KiB = 8 lines is due, PFD x 32 iterations ahead; wave-uniform address, the loaded dword is never read
            const uint32_t j = (uint32_t)it + 32u * PFD + 3u;
            const unsigned long long a = pb + (unsigned long long)((uint32_t)wave * q + ((j * 1024u) & (q - 1)));
            const uint32_t alo = __builtin_amdgcn_readfirstlane((uint32_t)a), ahi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
            const unsigned long long au = ((unsigned long long)ahi << 32) | alo;
            asm volatile("s_load_dword %0, %1, 0x0\n\ts_load_dword %0, %1, 0x80\n\ts_load_dword %0, %1, 0x100\n\ts_load_dword %0, %1, 0x180\n\t"
                         "s_load_dword %0, %1, 0x200\n\ts_load_dword %0, %1, 0x280\n\ts_load_dword %0, %1, 0x300\n\ts_load_dword %0, %1, 0x380"
                         : "+s"(ssink) : "s"(au) : "memory");
        }
        if (MIX && (it & 3) != 3) __builtin_amdgcn_raw_ptr_buffer_load_lds(rh, dst, 16, voff, (uint32_t)wave * (262144u / W) + (((uint32_t)it * 1024u) & (262144u / W - 1)), 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voff, soff, 0, 0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1 + (MIX == 2 ? 1 : 0)) : "memory");   // (the scalar prefetches count in lgkmcnt: nothing waits for them)
    }
    asm volatile("" ::"v"(sink));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    asm volatile("" ::"s"(ssink));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

struct Case { const char *name; size_t wg_stride; uint32_t span; int iters; bool rotate; int mix; };

template <int DEPTH, int W = 4, int PFD = 2>
static void run(const Case &c, const uint8_t *hot, const uint8_t *buf, size_t buf_bytes, unsigned long long *d_cyc, int blocks)
{
    std::vector<unsigned long long> h(blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f; double med = 0;
    const size_t need = c.wg_stride * blocks ? c.wg_stride * blocks : c.span;
    for (int r = 0; r < 5; ++r) {
        const uint8_t *base = buf + (c.rotate ? (size_t)(r % (int)(buf_bytes / need)) * need : 0);
        hipEventRecord(e0);
        const int iters = c.iters * 4 / W;
        if (c.mix == 3) hipLaunchKernelGGL((mlp<DEPTH, 3, W, PFD>), dim3(blocks), dim3(W * 64), 0, 0, hot, base, c.wg_stride, c.span, iters, d_cyc);
        else if (c.mix == 2) hipLaunchKernelGGL((mlp<DEPTH, 2, W, PFD>), dim3(blocks), dim3(W * 64), 0, 0, hot, base, c.wg_stride, c.span, iters, d_cyc);
        else if (c.mix) hipLaunchKernelGGL((mlp<DEPTH, 1, W, PFD>), dim3(blocks), dim3(W * 64), 0, 0, hot, base, c.wg_stride, c.span, iters, d_cyc);
        else hipLaunchKernelGGL((mlp<DEPTH, 0, W, PFD>), dim3(blocks), dim3(W * 64), 0, 0, hot, base, c.wg_stride, c.span, iters, d_cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r == 0) continue;   // warm-up (fills the cache level under test)
        if (ms < best) {
            best = ms;
            hipMemcpy(h.data(), d_cyc, blocks * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            med = (double)h[blocks / 2];
        }
    }
    const double bytes_wg = (double)W * (c.iters * 4 / W) * 1024.0;
    printf("%-34s %d waves, depth %2d (%4d lines in flight per CU)%s: %6.1f B/clk/CU (median CU, in-kernel cycles)  chip %6.2f TB/s  %8.1f us\n", c.name, W, DEPTH, W * DEPTH * 8,
           c.mix == 3 ? (PFD == 1 ? ", SCALAR prefetch 1 block ahead" : PFD == 2 ? ", SCALAR prefetch 2 blocks ahead" : ", SCALAR prefetch 4 blocks ahead") : c.mix == 2 ? (PFD == 1 ? ", prefetch 1 block ahead" : PFD == 2 ? ", prefetch 2 blocks ahead" : ", prefetch 4 blocks ahead") : "",
           bytes_wg / med, bytes_wg * blocks / (best * 1e-3) / 1e12, best * 1e3);
}

int main()
{
    const int blocks = 256;
    const size_t buf_bytes = (size_t)3 << 30;
    uint8_t *buf, *hot; unsigned long long *d_cyc;
    if (hipMalloc(&buf, buf_bytes) != hipSuccess || hipMalloc(&hot, 256 * 1024) != hipSuccess || hipMalloc(&d_cyc, blocks * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, buf_bytes); hipMemset(hot, 1, 256 * 1024);
    const Case cases[] = {
        {"L2 (256 KiB shared by all CUs)", 0, 256 * 1024, 512, false, 0},
        {"Infinity Cache (512 KiB per CU)", 512 * 1024, 512 * 1024, 128, false, 0},
        {"HBM (cold 4 MiB per CU)", (size_t)4 << 20, 4u << 20, 512, true, 0},
        {"3 : 1 L2 : HBM (tile-kernel mix)", (size_t)4 << 20, 4u << 20, 1024, true, 1},
    };
    const Case mixpf = {"3 : 1 L2 : HBM + L2 prefetch", (size_t)4 << 20, 4u << 20, 1024, true, 2};
    if (getenv("MLP_QUICK") == nullptr)
    for (const Case &c : cases) {
        run<1>(c, hot, buf, buf_bytes, d_cyc, blocks); run<2>(c, hot, buf, buf_bytes, d_cyc, blocks); run<4>(c, hot, buf, buf_bytes, d_cyc, blocks);
        run<8>(c, hot, buf, buf_bytes, d_cyc, blocks); run<16>(c, hot, buf, buf_bytes, d_cyc, blocks); run<32>(c, hot, buf, buf_bytes, d_cyc, blocks);
    }
    // the tile-kernel mix: two waves per SIMD, and with the cold lines prefetched ahead of the demand stream
    run<4, 8>(cases[3], hot, buf, buf_bytes, d_cyc, blocks); run<8, 8>(cases[3], hot, buf, buf_bytes, d_cyc, blocks); run<16, 8>(cases[3], hot, buf, buf_bytes, d_cyc, blocks);
    run<4, 4, 2>(mixpf, hot, buf, buf_bytes, d_cyc, blocks); run<8, 4, 2>(mixpf, hot, buf, buf_bytes, d_cyc, blocks); run<16, 4, 2>(mixpf, hot, buf, buf_bytes, d_cyc, blocks);
    run<8, 4, 1>(mixpf, hot, buf, buf_bytes, d_cyc, blocks); run<8, 4, 4>(mixpf, hot, buf, buf_bytes, d_cyc, blocks); run<16, 4, 4>(mixpf, hot, buf, buf_bytes, d_cyc, blocks);
    run<8, 8, 2>(mixpf, hot, buf, buf_bytes, d_cyc, blocks); run<16, 8, 2>(mixpf, hot, buf, buf_bytes, d_cyc, blocks);
    // the same, the cold lines touched through the scalar cache path (s_load_dword, one per 128-byte line)
    const Case mixsp = {"3 : 1 L2 : HBM + scalar prefetch", (size_t)4 << 20, 4u << 20, 1024, true, 3};
    run<4, 4, 2>(mixsp, hot, buf, buf_bytes, d_cyc, blocks); run<8, 4, 2>(mixsp, hot, buf, buf_bytes, d_cyc, blocks); run<16, 4, 2>(mixsp, hot, buf, buf_bytes, d_cyc, blocks);
    run<8, 4, 1>(mixsp, hot, buf, buf_bytes, d_cyc, blocks); run<8, 4, 4>(mixsp, hot, buf, buf_bytes, d_cyc, blocks); run<16, 4, 4>(mixsp, hot, buf, buf_bytes, d_cyc, blocks);
    run<8, 8, 2>(mixsp, hot, buf, buf_bytes, d_cyc, blocks); run<16, 8, 2>(mixsp, hot, buf, buf_bytes, d_cyc, blocks);
    return 0;
}
