// Measured FLOORS of the tile kernels behind bench.py's headline workloads (NOT part of the product library).
//
// This file compiles the product's ring-kernel source (fp8-mps-metal_amd/csrc/fp8mi_gemm.hip) a second time with FP8MI_FLOOR_PROBE, which
// adds three timing-only instantiations per kernel (Cfg::FLOOR, wrong results by construction):
//     1  the K loop's LDS-DMA stream with its waits and barriers, no fragment reads, no MFMAs      -> dma_only_us
//     2  no K loop: launch, arguments, tile map, the fused epilogue's C store, kernel end          -> c_store_only_us
//     3  return behind the argument loads: the launch itself (same grid, block, LDS allocation)    -> empty_launch_us
// bench.py times them in the run, with per-dispatch start/stop events exactly as it times the real kernel, on the same rotating weight
// buffers, and prints them as `roofline.floor` next to the kernel's own time: "0.22 of spec" can then be read next to what ONE dispatch of
// this shape costs on this box before it multiplies anything.  Everything is hidden-visibility except the C entry points below, so nothing
// here can shadow a symbol of libfp8mi.so in the same process.
//   hipcc --offload-arch=gfx950 -O3 -fPIC -shared -fvisibility=hidden tools/floor_probe.hip -o tools/libfloor_probe.so
#define FP8MI_FLOOR_PROBE 1
#include "../fp8-mps-metal_amd/csrc/fp8mi_gemm.hip"

#include <vector>

// --- what fp8mi_gemm.hip expects from the rest of the library (fp8mi_api.hip / fp8mi_gemm256.hip), local to this probe -------------------
namespace {
struct Events {
    std::vector<hipEvent_t> ev;
    int used = 0;
    bool on = false;
};
Events g_ev;
}  // namespace

bool fp8mi_next_profile_events(hipEvent_t *start, hipEvent_t *stop)
{
    if (!g_ev.on || (size_t)(2 * g_ev.used + 1) >= g_ev.ev.size()) return false;
    *start = g_ev.ev[2 * g_ev.used];
    *stop = g_ev.ev[2 * g_ev.used + 1];
    ++g_ev.used;
    return true;
}

int fp8mi_cu_count()
{
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
}
int fp8mi_launch_gemm256(const MMParams &, int, hipStream_t) { return FP8MI_E_UNSUPPORTED; }
bool fp8mi_gemm256_supported(const MMParams &) { return false; }
// (fp8mi_gemm.hip's automatic tile choice goes through the dispatch's cost model, which asks the other kernels' envelopes; the probe only launches forced ids)
bool fp8mi_gemv_supported(const MMParams &) { return false; }
bool fp8mi_gemv_mx_supported(const MMParams &) { return false; }
bool fp8mi_skinny_supported(const MMParams &) { return false; }

extern "C" {

__attribute__((visibility("default"))) int floor_probe_abi_version() { return 1; }

// Runs `reps` launches of floor form `id` (901-903: the 128x64 tile kernel of config C3; 911-913: the deep-ring 128x128 kernel) over the
// `nb` weight buffers in Bs (rotating), each launch with its own start/stop event pair; writes the per-launch times (ms) to ms_out[reps].
// per-tensor scales, no bias, out_dtype as given.  Returns 0, a hipError_t, or a negative FP8MI_E_* code.
__attribute__((visibility("default"))) int floor_probe_run(int id, const void *A, const void *const *Bs, int nb, void *C, const float *sa, const float *sb,
                                                           long long M, long long N, long long K, int out_dtype, int reps, void *stream, float *ms_out)
{
    if (reps <= 0 || nb <= 0 || !ms_out) return FP8MI_E_SHAPE;
    while (g_ev.ev.size() < (size_t)reps * 2) {
        hipEvent_t e;
        hipError_t rc = hipEventCreate(&e);
        if (rc != hipSuccess) return (int)rc;
        g_ev.ev.push_back(e);
    }
    MMParams p = {};
    p.A = (const uint8_t *)A; p.C = C; p.scale_a = sa; p.scale_b = sb;
    p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N;
    p.out_dtype = out_dtype; p.nan_zero = 1; p.split = 1;
    g_ev.used = 0;
    g_ev.on = true;
    int rc = 0;
    for (int i = 0; i < reps && rc == 0; ++i) {
        p.B = (const uint8_t *)Bs[i % nb];
        rc = fp8mi_launch_gemm(p, id, (hipStream_t)stream);
    }
    g_ev.on = false;
    if (rc) return rc;
    const int n = g_ev.used;
    if (n != reps) return FP8MI_E_UNSUPPORTED;
    hipError_t e = hipEventSynchronize(g_ev.ev[2 * (n - 1) + 1]);
    if (e != hipSuccess) return (int)e;
    for (int i = 0; i < n; ++i) {
        e = hipEventElapsedTime(&ms_out[i], g_ev.ev[2 * i], g_ev.ev[2 * i + 1]);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

}  // extern "C"
