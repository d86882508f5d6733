/*
 * Torch-free use of the peer-store all-gather (include/fp8mi_peer.h): a plain C host program, one process per rank, no Python and no
 * torch.distributed anywhere - what a cgo / JNI host would do.  The parent forks WORLD children BEFORE anything touches HIP (the parent
 * itself never does); the children exchange their 64-byte IPC handles through an anonymous shared mapping, map each other's buffers, and run
 * ROUNDS back-to-back gathers of two slab sizes with a pattern per (rank, round), each checked on the host; then one rank makes a call nobody
 * joins and must get the timeout status bits instead of a hung GPU.  All ranks share device 0 (HIP IPC maps another process's allocation
 * whichever device it lives on).  Built and run by tests/test_gpu_c_abi.py:
 *   gcc -D__HIP_PLATFORM_AMD__ tests/c/peer_roundtrip.c -I/opt/rocm/include -Iinclude -Lfp8-mps-metal_amd -lfp8mi_peer -L/opt/rocm/lib -lamdhip64 -o ...
 * Exit code 0 = every rank passed every check.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include "fp8mi_peer.h"

#define WORLD 3
#define ROUNDS 12
#define SLAB (3 * 4096 + 16)

struct shared {
    volatile int arrived[8];           /* one counter per barrier use */
    unsigned char data_handle[WORLD][FP8MI_PEER_HANDLE_BYTES];
    unsigned char flag_handle[WORLD][FP8MI_PEER_HANDLE_BYTES];
};

static void barrier(struct shared *sh, int which)
{
    __sync_fetch_and_add(&sh->arrived[which], 1);
    for (int spins = 0; sh->arrived[which] < WORLD; ++spins) {
        if (spins > 600000) { printf("host barrier %d timed out\n", which); _exit(9); }   /* 60 s */
        usleep(100);
    }
}

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("rank %d: HIP error %d at line %d\n", rank, (int)e_, __LINE__); return 2; } } while (0)
#define CHECK_PEER(x) do { int r_ = (x); if (r_ != 0) { printf("rank %d: fp8mi_peer error %d: %s (line %d)\n", rank, r_, fp8mi_peer_last_error(), __LINE__); return 3; } } while (0)

static unsigned char pattern(int rank, int round, int i) { return (unsigned char)((rank * 37 + round * 11 + i) % 251); }

static int run_rank(int rank, struct shared *sh)
{
    CHECK_HIP(hipSetDevice(0));
    void *data = NULL, *flags = NULL;
    const int64_t nbytes = (int64_t)WORLD * SLAB;
    CHECK_PEER(fp8mi_peer_alloc(nbytes, 0, &data));
    CHECK_PEER(fp8mi_peer_alloc(0, 1, &flags));
    CHECK_PEER(fp8mi_peer_export(data, sh->data_handle[rank]));
    CHECK_PEER(fp8mi_peer_export(flags, sh->flag_handle[rank]));
    barrier(sh, 0);                                       /* every handle is published */
    void *dptr[WORLD], *fptr[WORLD];
    for (int r = 0; r < WORLD; ++r) {
        if (r == rank) { dptr[r] = data; fptr[r] = flags; continue; }
        CHECK_PEER(fp8mi_peer_open(sh->data_handle[r], &dptr[r]));
        CHECK_PEER(fp8mi_peer_open(sh->flag_handle[r], &fptr[r]));
    }
    fp8mi_peer_ctx *ctx = NULL;
    CHECK_PEER(fp8mi_peer_ctx_create(WORLD, rank, dptr, fptr, nbytes, &ctx));
    barrier(sh, 1);                                       /* everybody has mapped everybody */

    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    unsigned char *host = NULL, *snap = NULL;             /* pinned: the copies below stay asynchronous */
    CHECK_HIP(hipHostMalloc((void **)&host, (size_t)SLAB * ROUNDS, 0));
    CHECK_HIP(hipHostMalloc((void **)&snap, (size_t)nbytes * ROUNDS, 0));
    for (int round = 0; round < ROUNDS; ++round) {        /* back to back: no host synchronisation between the rounds */
        const int n = round % 2 == 0 ? SLAB : 4096;
        unsigned char *src = host + (size_t)round * SLAB;
        for (int i = 0; i < n; ++i) src[i] = pattern(rank, round, i);
        CHECK_HIP(hipMemcpyAsync((char *)data + (size_t)rank * SLAB, src, (size_t)n, hipMemcpyHostToDevice, stream));
        CHECK_PEER(fp8mi_peer_allgather(ctx, (int64_t)rank * SLAB, n, 20 * 1000 * 1000, stream));
        CHECK_HIP(hipMemcpyAsync(snap + (size_t)round * nbytes, data, (size_t)nbytes, hipMemcpyDeviceToHost, stream));   /* the consumer: stream-ordered behind the gather */
    }
    uint32_t status = 77;
    CHECK_PEER(fp8mi_peer_status(ctx, stream, &status));
    if (status != 0) { printf("rank %d: status %u after the rounds\n", rank, status); return 4; }
    for (int round = 0; round < ROUNDS; ++round) {
        const int n = round % 2 == 0 ? SLAB : 4096;
        for (int r = 0; r < WORLD; ++r)
            for (int i = 0; i < n; ++i)
                if (snap[(size_t)round * nbytes + (size_t)r * SLAB + i] != pattern(r, round, i)) {
                    printf("rank %d: round %d, slab of rank %d, byte %d wrong\n", rank, round, r, i);
                    return 5;
                }
    }
    /* argument errors are errors */
    if (fp8mi_peer_allgather(ctx, 8, 16, 0, stream) != -2 || fp8mi_peer_allgather(ctx, nbytes, 16, 0, stream) != -2) { printf("rank %d: bad slab accepted\n", rank); return 6; }
    barrier(sh, 2);
    /* a call nobody joins: bounded waits, status bits, no hang */
    if (rank == 0) {
        CHECK_PEER(fp8mi_peer_allgather(ctx, 0, 1024, 300 * 1000, stream));
        CHECK_PEER(fp8mi_peer_status(ctx, stream, &status));
        if (status != (FP8MI_PEER_TIMEOUT_READY | FP8MI_PEER_TIMEOUT_DONE)) { printf("rank 0: timeout status %u\n", status); return 7; }
    }
    CHECK_HIP(hipDeviceSynchronize());
    barrier(sh, 3);                                       /* nobody stores into anybody any more */
    for (int r = 0; r < WORLD; ++r)
        if (r != rank) { CHECK_PEER(fp8mi_peer_close(dptr[r])); CHECK_PEER(fp8mi_peer_close(fptr[r])); }
    barrier(sh, 4);                                       /* ... and nobody maps anybody */
    CHECK_PEER(fp8mi_peer_ctx_destroy(ctx));
    CHECK_PEER(fp8mi_peer_free(data));
    CHECK_PEER(fp8mi_peer_free(flags));
    CHECK_HIP(hipHostFree(host)); CHECK_HIP(hipHostFree(snap));
    return 0;
}

int main(void)
{
    struct shared *sh = mmap(NULL, sizeof(struct shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (sh == MAP_FAILED) { printf("mmap failed\n"); return 1; }
    memset((void *)sh, 0, sizeof(*sh));
    pid_t pids[WORLD];
    for (int r = 0; r < WORLD; ++r) {
        pids[r] = fork();                                 /* before any HIP call: every rank initialises the GPU itself */
        if (pids[r] < 0) { printf("fork failed\n"); return 1; }
        if (pids[r] == 0) { const int rc = run_rank(r, sh); fflush(stdout); _exit(rc); }
    }
    int bad = 0;
    for (int r = 0; r < WORLD; ++r) {
        int st = 0;
        waitpid(pids[r], &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { printf("rank %d failed (status %d)\n", r, st); bad = 1; }
    }
    if (!bad) printf("peer all-gather C round trip: ok (%d ranks, %d rounds)\n", WORLD, ROUNDS);
    return bad;
}
