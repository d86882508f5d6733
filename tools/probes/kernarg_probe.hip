// What does a kernel pay for fetching its arguments, and does kernarg PRELOAD (the dispatcher places the first dwords of the kernarg segment in
// SGPRs before the first wave starts: -mllvm -amdgpu-kernarg-preload-count=N, gfx950) remove it?  Two kernels with the same body - one dependent
// global load through a pointer argument, one store - launched as 256 workgroups of 256 threads, timed per dispatch (start / stop events of
// hipExtLaunchKernel, as the product times its kernels):
//   byref   : the arguments arrive in a by-value struct (what MMParams is): the kernel starts with s_load + wait
//   preload : the same values as leading scalar arguments: preloaded into SGPRs, no load before the first use
// Run it with HIP_FORCE_DEV_KERNARG=1 and =0 (kernarg segment in device / host memory).
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 kernarg_probe.hip -o kernarg_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct Args { const float *src; float *dst; long n; long pad[18]; };   // 168 bytes, the size of MMParams

__global__ __launch_bounds__(256) void k_byref(Args a)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < a.n) a.dst[i] = a.src[i] + (float)a.pad[17];
}
__global__ __launch_bounds__(256) void k_preload(const float *src, float *dst, long n, Args rest)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i] + (float)rest.pad[17];
}
__global__ __launch_bounds__(256) void k_empty(Args a) { if (a.n < 0) a.dst[0] = 0; }
__global__ __launch_bounds__(256) void k_empty_preload(const float *src, float *dst, long n) { if (n < 0) dst[0] = 0; }

template <typename... A>
static double time_kernel(void (*k)(A...), int reps, A... args)
{
    void *argv[] = {(void *)&args...};
    std::vector<hipEvent_t> ev(2 * reps);
    for (auto &e : ev) hipEventCreate(&e);
    for (int i = 0; i < 20; ++i) hipLaunchKernel((const void *)k, dim3(256), dim3(256), argv, 0, 0);
    hipDeviceSynchronize();
    for (int i = 0; i < reps; ++i) hipExtLaunchKernel((const void *)k, dim3(256), dim3(256), argv, 0, 0, ev[2 * i], ev[2 * i + 1], 0);
    hipDeviceSynchronize();
    std::vector<float> ms(reps);
    for (int i = 0; i < reps; ++i) hipEventElapsedTime(&ms[i], ev[2 * i], ev[2 * i + 1]);
    std::sort(ms.begin(), ms.end());
    double avg = 0; for (float m : ms) avg += m;
    for (auto &e : ev) hipEventDestroy(e);
    printf("  avg %6.2f us  median %6.2f  min %6.2f\n", avg / reps * 1e3, ms[reps / 2] * 1e3, ms[0] * 1e3);
    return avg / reps * 1e3;
}

int main()
{
    const char *e = getenv("HIP_FORCE_DEV_KERNARG");
    printf("HIP_FORCE_DEV_KERNARG=%s\n", e ? e : "(unset)");
    float *src, *dst; hipMalloc(&src, 65536 * 4); hipMalloc(&dst, 65536 * 4); hipMemset(src, 0, 65536 * 4);
    Args a = {}; a.src = src; a.dst = dst; a.n = 65536;
    for (int round = 0; round < 2; ++round) {
        printf("empty, byref struct  :"); time_kernel(k_empty, 400, a);
        printf("empty, preload       :"); time_kernel(k_empty_preload, 400, (const float *)src, dst, (long)65536);
        printf("load+store, byref    :"); time_kernel(k_byref, 400, a);
        printf("load+store, preload  :"); time_kernel(k_preload, 400, (const float *)src, dst, (long)65536, a);
    }
    return 0;
}
