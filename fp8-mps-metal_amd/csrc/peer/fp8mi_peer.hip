// libfp8mi_peer.so: the direct (peer-store) all-gather of the sharded linear's output - see include/fp8mi_peer.h for
// the contract.  Two small kernels per call; every wait is bounded by the device's wall clock.
//
// Memory model notes (gfx950, 8 XCDs with private L2s):
//   * flag blocks are UNCACHED device memory (hipDeviceMallocUncached): a peer's store is visible to the owner's next
//     load and the other way round, no L2 in between; flags are read and written with system-scope atomics;
//   * the slab is stored with `sc0 sc1` (system scope: written through the storing XCD's L2), every wave drains its
//     stores before it ends, and the `done` flag is only written by the NEXT kernel on the stream - the end of the push
//     kernel has made its stores visible device-wide and beyond by then;
//   * the receiver's consumers are kernels enqueued behind the `end` kernel: a kernel start invalidates the L2 lines an
//     earlier kernel may have left of the buffer (what makes a producer kernel on one XCD visible to a consumer kernel on
//     another is the same mechanism), so they read what the peers stored.

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../../include/fp8mi.h"
#include "../../../include/fp8mi_peer.h"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kMax = FP8MI_PEER_MAX_RANKS;
// flag block, in 32-bit words
constexpr int kReady = 0;          // [kMax] ready[i]: rank i may be sent the slab of epoch <= value        (written by rank i)
constexpr int kDone = kMax;        // [kMax] done[i] : rank i's slab of epoch <= value has landed here       (written by rank i)
constexpr int kEpoch = 2 * kMax;   // this rank's epoch counter                                             (local)
constexpr int kStatus = 2 * kMax + 1;  // FP8MI_PEER_TIMEOUT_* bits                                         (local)

struct PeerTab {
    uint8_t *data[kMax];
    uint32_t *flags[kMax];
    int world, rank;
};

thread_local char g_err[256] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(call)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) return fail((int)e_, "%s: %s", #call, hipGetErrorString(e_));            \
    } while (0)

__device__ __forceinline__ uint32_t flag_load(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void flag_store(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }

// spin until *p has reached epoch e (wrap-safe compare) or `ticks` of the wall clock have passed; true = reached
__device__ bool wait_flag(const uint32_t *p, uint32_t e, int64_t ticks)
{
    const uint64_t t0 = wall_clock64();
    for (;;) {
        if ((int32_t)(flag_load(p) - e) >= 0) {
            __atomic_thread_fence(__ATOMIC_ACQUIRE);   // (system scope by default: what follows sees what preceded the flag's release)
            return true;
        }
        if ((int64_t)(wall_clock64() - t0) > ticks) return false;
        __builtin_amdgcn_s_sleep(16);
    }
}

// The epoch of a call is (own counter + 1); the counter is device-resident (kernel arguments would be frozen by a graph capture) and is advanced
// by the call's `end` kernel, so both kernels of a call read the same value.
//
// push: blockIdx.y picks the peer (rank r starts with r+1: at every moment the ranks store to different peers), blockIdx.x a share of the slab.
// Block (0, y) first tells its peer "rank r is ready to receive epoch e" - this kernel is behind the rank's consumers of the previous gather on
// the stream - then every block of the row waits for the same word from the peer.  (With bytes == 0 the grid is (1, world-1): the handshake only.)
__global__ __launch_bounds__(256) void peer_push_kernel(PeerTab t, int64_t offset, int64_t bytes, int64_t ticks)
{
    __shared__ int go;
    uint32_t *mine = t.flags[t.rank];
    const int p = (t.rank + 1 + (int)blockIdx.y) % t.world;
    if (threadIdx.x == 0) {
        const uint32_t e = flag_load(mine + kEpoch) + 1u;
        if (blockIdx.x == 0) flag_store(t.flags[p] + kReady + t.rank, e);
        go = wait_flag(mine + kReady + p, e, ticks);
        if (!go) atomicOr(mine + kStatus, FP8MI_PEER_TIMEOUT_READY);
    }
    __syncthreads();
    if (!go) return;
    const u32x4 *src = (const u32x4 *)(t.data[t.rank] + offset);
    uint8_t *dst = t.data[p] + offset;
    const int64_t pieces = bytes >> 4;
    // kDepth 16-byte loads in flight per lane, then their stores (posted: nothing waits for them until the wave ends).  Few, deep blocks on purpose: the
    // push shares the GPU with the NEXT chunk's GEMM, whose one-wave-per-SIMD tiles need a CU's whole register file - a push wave resident on a
    // CU keeps such a tile off it, so the copy is confined to a few dozen CUs and gets its bytes in flight from depth, not from width
    constexpr int kDepth = 16;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pieces; i += kDepth * stride) {
        u32x4 v[kDepth];
#pragma unroll
        for (int u = 0; u < kDepth; ++u)
            if (i + u * stride < pieces) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < kDepth; ++u)
            if (i + u * stride < pieces) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst + ((i + u * stride) << 4)), "v"(v[u]) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// end: this rank's slab has landed everywhere (the push kernel precedes this one on the stream); wait for everybody else's; advance the epoch
__global__ __launch_bounds__(64) void peer_end_kernel(PeerTab t, int64_t ticks)
{
    uint32_t *mine = t.flags[t.rank];
    const uint32_t e = flag_load(mine + kEpoch) + 1u;   // one wave: every lane has read the counter before lane 0 replaces it below
    const int p = threadIdx.x;
    if (p < t.world && p != t.rank) {
        flag_store(t.flags[p] + kDone + t.rank, e);
        if (!wait_flag(mine + kDone + p, e, ticks)) atomicOr(mine + kStatus, FP8MI_PEER_TIMEOUT_DONE);
    }
    if (p == 0) flag_store(mine + kEpoch, e);
}

}  // namespace

struct fp8mi_peer_ctx {
    PeerTab tab;
    int64_t data_bytes;
    int64_t wall_khz;
    int push_blocks;   // per peer; 0 = by size
};

extern "C" {

int fp8mi_peer_version(void) { return FP8MI_PEER_VERSION; }
const char *fp8mi_peer_last_error(void) { return g_err; }

int fp8mi_peer_alloc(int64_t bytes, int flag_block, void **ptr_out)
{
    if (!ptr_out) return fail(FP8MI_E_NULL, "fp8mi_peer_alloc: ptr_out is NULL");
    if (flag_block) bytes = FP8MI_PEER_FLAG_BYTES;
    if (bytes <= 0) return fail(FP8MI_E_SHAPE, "fp8mi_peer_alloc: bytes = %lld", (long long)bytes);
    void *p = nullptr;
    if (flag_block) HIP_TRY(hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached));
    else HIP_TRY(hipMalloc(&p, (size_t)bytes));
    hipError_t e = hipMemset(p, 0, (size_t)bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(p);
        return fail((int)e, "fp8mi_peer_alloc: zero fill: %s", hipGetErrorString(e));
    }
    *ptr_out = p;
    return 0;
}

int fp8mi_peer_free(void *ptr)
{
    if (!ptr) return 0;
    HIP_TRY(hipFree(ptr));
    return 0;
}

int fp8mi_peer_export(void *ptr, void *handle_out)
{
    if (!ptr || !handle_out) return fail(FP8MI_E_NULL, "fp8mi_peer_export: NULL argument");
    static_assert(sizeof(hipIpcMemHandle_t) == FP8MI_PEER_HANDLE_BYTES, "handle size");
    hipIpcMemHandle_t h;
    HIP_TRY(hipIpcGetMemHandle(&h, ptr));
    memcpy(handle_out, &h, sizeof(h));
    return 0;
}

int fp8mi_peer_open(const void *handle, void **ptr_out)
{
    if (!handle || !ptr_out) return fail(FP8MI_E_NULL, "fp8mi_peer_open: NULL argument");
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void *p = nullptr;
    HIP_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    *ptr_out = p;
    return 0;
}

int fp8mi_peer_close(void *ptr)
{
    if (!ptr) return 0;
    HIP_TRY(hipIpcCloseMemHandle(ptr));
    return 0;
}

int fp8mi_peer_ctx_create(int world, int rank, void *const *data_ptrs, void *const *flag_ptrs, int64_t data_bytes, fp8mi_peer_ctx **ctx_out)
{
    if (!data_ptrs || !flag_ptrs || !ctx_out) return fail(FP8MI_E_NULL, "fp8mi_peer_ctx_create: NULL argument");
    if (world < 2 || world > kMax || rank < 0 || rank >= world) return fail(FP8MI_E_SHAPE, "fp8mi_peer_ctx_create: world = %d (2..%d), rank = %d", world, kMax, rank);
    if (data_bytes <= 0 || data_bytes % 16) return fail(FP8MI_E_SHAPE, "fp8mi_peer_ctx_create: data_bytes = %lld (a positive multiple of 16)", (long long)data_bytes);
    fp8mi_peer_ctx *c = (fp8mi_peer_ctx *)calloc(1, sizeof(fp8mi_peer_ctx));
    if (!c) return fail(FP8MI_E_NULL, "fp8mi_peer_ctx_create: out of host memory");
    for (int i = 0; i < world; ++i) {
        if (!data_ptrs[i] || !flag_ptrs[i] || ((uintptr_t)data_ptrs[i] & 15)) {
            free(c);
            return fail(FP8MI_E_NULL, "fp8mi_peer_ctx_create: rank %d's buffer or flag block is NULL or not 16-byte aligned", i);
        }
        c->tab.data[i] = (uint8_t *)data_ptrs[i];
        c->tab.flags[i] = (uint32_t *)flag_ptrs[i];
    }
    c->tab.world = world;
    c->tab.rank = rank;
    c->data_bytes = data_bytes;
    int dev = 0, khz = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev);
    if (e != hipSuccess || khz <= 0) khz = 100000;   // gfx9: a 100 MHz constant clock
    c->wall_khz = khz;
    const char *pb = getenv("FP8MI_PEER_BLOCKS");
    c->push_blocks = pb ? atoi(pb) : 0;
    *ctx_out = c;
    return 0;
}

int fp8mi_peer_ctx_destroy(fp8mi_peer_ctx *ctx)
{
    free(ctx);
    return 0;
}

int fp8mi_peer_allgather(fp8mi_peer_ctx *ctx, int64_t offset, int64_t bytes, int64_t timeout_us, void *stream)
{
    if (!ctx) return fail(FP8MI_E_NULL, "fp8mi_peer_allgather: ctx is NULL");
    if (offset < 0 || bytes < 0 || (offset & 15) || (bytes & 15) || offset + bytes > ctx->data_bytes)
        return fail(FP8MI_E_SHAPE, "fp8mi_peer_allgather: slab [%lld, +%lld) must be 16-byte granular and inside the %lld-byte buffer",
                    (long long)offset, (long long)bytes, (long long)ctx->data_bytes);
    if (timeout_us <= 0) timeout_us = 30ll * 1000 * 1000;
    const int64_t ticks = timeout_us * ctx->wall_khz / 1000;
    hipStream_t s = (hipStream_t)stream;
    {
        // blocks per peer: one per 512 KiB of slab, at most 12 (FP8MI_PEER_BLOCKS overrides, up to 64).  One block moves ~45 GB/s of same-device copy
        // (tools/time_peer_gather.py, profiles/r04_peer_rehearsal.txt); a link takes 153 GB/s; 7 peers x 12 blocks leave two thirds of the CUs to the GEMM
        int bx = ctx->push_blocks > 0 ? ctx->push_blocks : (int)((bytes + (1 << 19) - 1) >> 19);
        const int cap = ctx->push_blocks > 0 ? 64 : 12;
        bx = (bx < 1 || bytes == 0) ? 1 : bx > cap ? cap : bx;
        hipLaunchKernelGGL(peer_push_kernel, dim3(bx, ctx->tab.world - 1), dim3(256), 0, s, ctx->tab, offset, bytes, ticks);
    }
    hipLaunchKernelGGL(peer_end_kernel, dim3(1), dim3(64), 0, s, ctx->tab, ticks);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "fp8mi_peer_allgather: launch: %s", hipGetErrorString(e));
    return 0;
}

int fp8mi_peer_status(fp8mi_peer_ctx *ctx, void *stream, uint32_t *status_out)
{
    if (!ctx || !status_out) return fail(FP8MI_E_NULL, "fp8mi_peer_status: NULL argument");
    hipStream_t s = (hipStream_t)stream;
    uint32_t *word = ctx->tab.flags[ctx->tab.rank] + kStatus;
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipMemcpy(status_out, word, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (*status_out) HIP_TRY(hipMemset(word, 0, sizeof(uint32_t)));
    return 0;
}

}  // extern "C"
