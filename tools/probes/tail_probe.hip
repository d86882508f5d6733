// Cycle cost of the fused tail's instruction classes, one wave per SIMD (256 threads, 512-register launch bounds are not needed here):
// each variant repeats a block of 32 "items" and reports cycles per item.   hipcc --offload-arch=gfx950 -O3 tail_probe.hip -o tail_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define REP32(x) x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x x
#define READS "v_accvgpr_read_b32 v128, a0\n\tv_accvgpr_read_b32 v129, a1\n\tv_accvgpr_read_b32 v130, a2\n\tv_accvgpr_read_b32 v131, a3\n\tv_accvgpr_read_b32 v132, a4\n\tv_accvgpr_read_b32 v133, a5\n\tv_accvgpr_read_b32 v134, a6\n\tv_accvgpr_read_b32 v135, a7\n\t"
#define SWAPS "v_permlane16_swap_b32 v128, v132\n\tv_permlane16_swap_b32 v129, v133\n\tv_permlane16_swap_b32 v130, v134\n\tv_permlane16_swap_b32 v131, v135\n\t"
#define MULS "v_pk_mul_f32 v[128:129], v[128:129], v[120:121]\n\tv_pk_mul_f32 v[130:131], v[130:131], v[120:121]\n\tv_pk_mul_f32 v[132:133], v[132:133], v[120:121]\n\tv_pk_mul_f32 v[134:135], v[134:135], v[120:121]\n\t" \
             "v_pk_mul_f32 v[128:129], v[128:129], v[122:123]\n\tv_pk_mul_f32 v[130:131], v[130:131], v[122:123]\n\tv_pk_mul_f32 v[132:133], v[132:133], v[122:123]\n\tv_pk_mul_f32 v[134:135], v[134:135], v[122:123]\n\t"
#define CVTS "v_cvt_pk_bf16_f32 v128, v128, v129\n\tv_cvt_pk_bf16_f32 v129, v130, v131\n\tv_cvt_pk_bf16_f32 v130, v132, v133\n\tv_cvt_pk_bf16_f32 v131, v134, v135\n\t"
#define STORE "buffer_store_dwordx4 v[128:131], %[voff], %[rc], %[t0] offen nt\n\ts_add_u32 %[t0], %[t0], %[srow]\n\t"
#define STORE_PLAIN "buffer_store_dwordx4 v[128:131], %[voff], %[rc], %[t0] offen\n\ts_add_u32 %[t0], %[t0], %[srow]\n\t"
#define MFMA "v_mfma_scale_f32_16x16x128_f8f6f4 a[8:11], v[136:143], v[144:151], a[8:11], v120, v120 op_sel_hi:[0,0,0]\n\t"
#define MFMA2 "v_mfma_scale_f32_16x16x128_f8f6f4 a[12:15], v[136:143], v[144:151], a[12:15], v120, v120 op_sel_hi:[0,0,0]\n\t"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(256) void probe(unsigned long long *out, unsigned char *C, unsigned ldc_b, int contiguous)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    // wave tile 128 rows x 256 B (bf16 x 128 cols) at C + block*... ; scattered form: lane -> row fr, 16 B at (fg&1)*32 + (fg>>1)*16; contiguous: 4 rows x 256 B
    unsigned voff = contiguous ? (unsigned)((lane >> 4) * ldc_b + (lane & 15) * 16) : (unsigned)(fr * ldc_b + ((fg & 1) * 32 + (fg >> 1) * 16));
    unsigned long long base = (unsigned long long)(C + ((size_t)blockIdx.x * 256 + (wave & 1) * 128) * ldc_b + (wave >> 1) * 256);
    u32x4 rc = {(unsigned)base, (unsigned)(base >> 32) & 0xFFFFu, 128u * ldc_b, 0x00020000u};
    rc[0] = __builtin_amdgcn_readfirstlane(rc[0]); rc[1] = __builtin_amdgcn_readfirstlane(rc[1]); rc[2] = __builtin_amdgcn_readfirstlane(rc[2]);
    unsigned srow = contiguous ? 4 * ldc_b : 0, t0 = 0;   // scattered form rewrites the same 16 rows (timing only)
    unsigned long long c0, c1;
    asm volatile("v_mov_b32 v120, 0x7f7f7f7f\n\tv_mov_b32 v121, 1.0\n\tv_mov_b32 v122, 1.0\n\tv_mov_b32 v123, 1.0\n\t" ::: "v120", "v121", "v122", "v123");
#define BODY(X) asm volatile("s_memtime %[c0]\n\ts_waitcnt lgkmcnt(0)\n\t" REP32(X) "s_waitcnt vmcnt(0)\n\ts_memtime %[c1]\n\ts_waitcnt lgkmcnt(0)\n\t" \
        : [c0] "=&s"(c0), [c1] "=&s"(c1), [t0] "+s"(t0) : [voff] "v"(voff), [rc] "s"(rc), [srow] "s"(srow) \
        : "memory", "scc", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", \
          "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15")
    if (V == 0) BODY(READS);
    if (V == 1) BODY(READS SWAPS);
    if (V == 2) BODY(READS SWAPS MULS);
    if (V == 3) BODY(READS SWAPS MULS CVTS);
    if (V == 4) BODY(READS SWAPS MULS CVTS STORE);
    if (V == 5) BODY(STORE);
    if (V == 6) BODY(STORE_PLAIN);
    if (V == 7) BODY(MFMA MFMA2);
    if (V == 8) BODY(MFMA READS MFMA2 SWAPS MULS CVTS);
    if (V == 9) BODY(MFMA READS MFMA2 SWAPS MULS CVTS STORE);
    if (V == 10) BODY(SWAPS);
    if (V == 11) BODY(MULS);
    if (V == 12) BODY(CVTS);
    if (lane == 0) out[blockIdx.x * 4 + wave] = c1 - c0;
}

int main(int argc, char **argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 256;   // workgroups (one per CU up to 256): fewer = fewer CUs storing at the same time
    const unsigned ldc_b = 12288 * 2;
    unsigned long long *d; unsigned char *C;
    hipMalloc(&d, blocks * 4 * 8); hipMalloc(&C, (size_t)256 * 256 * ldc_b);
    std::vector<unsigned long long> h(blocks * 4);
    const char *names[] = {"8 accvgpr_read", "+ 4 permlane16_swap", "+ 8 pk_mul", "+ 4 cvt_pk", "+ store nt (scattered 16 rows x 64 B)", "store nt only", "store plain only",
                           "2 MFMA only", "2 MFMA + VALU item", "2 MFMA + VALU item + store", "4 swaps only", "8 pk_mul only", "4 cvt only"};
    for (int contiguous = 0; contiguous < 2; ++contiguous)
    for (int v = 0; v < 13; ++v) {
        if (contiguous && !(v == 4 || v == 5 || v == 6 || v == 9)) continue;
        for (int r = 0; r < 3; ++r) {
            switch (v) {
#define C_(n) case n: hipLaunchKernelGGL(probe<n>, dim3(blocks), dim3(256), 0, 0, d, C, ldc_b, contiguous); break;
                C_(0) C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8) C_(9) C_(10) C_(11) C_(12)
            }
        }
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, blocks * 4 * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto x : h) s += x;
        printf("%-48s %s: %7.1f cycles per item (32 items, mean over %d waves)\n", names[v], contiguous ? "[stores: 4 rows x 256 B]" : "", s / h.size() / 32.0, (int)h.size());
    }
    return 0;
}
