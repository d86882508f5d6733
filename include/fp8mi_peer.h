/*
 * fp8mi_peer - direct (peer-store) all-gather of the sharded linear's output over xGMI, C ABI.
 *
 * Companion of include/fp8mi.h for the ONE exchange step the path has (BASELINE.json configs[3]: the
 * (M, N/8) blocks of C gathered on every GPU).  The reference has no counterpart - one Apple GPU, one command queue
 * (fp8_bridge.cpp:67) - so there is no reference interface to cite; what it replaces inside this build is the
 * `dist.all_gather_into_tensor` call of fp8-mps-metal_amd/fp8_sharded_linear.py.  xGMI on MI355X is point-to-point
 * (7 links per GPU): a ring collective is bound by ONE link per step, a direct all-gather has every rank store its slab
 * into all 7 peers at once and is bound by the 7 links together.  This library is that direct form:
 *
 *   - every rank owns one DATA buffer (the gather buffer, same size on every rank) and one small FLAG block
 *     (uncached device memory), both allocated here and exported as HIP IPC handles; the host exchanges the
 *     handles (64 bytes each) any way it likes (the Python host uses torch.distributed's object all-gather) and
 *     opens the peers';
 *   - fp8mi_peer_allgather enqueues two small kernels on the caller's stream (the call's epoch e = own counter + 1; the
 *     counter is device-resident, so a captured HIP graph replays correctly):
 *       push  : for each peer p (a different first peer on every rank): tell p "rank r is ready to receive epoch e"
 *               (the rank's earlier consumers precede this kernel in stream order), wait until p is ready for e, then
 *               store this rank's slab [offset, offset+bytes) of its own buffer to the same range of p's buffer
 *               (system-scope, written through);
 *       end   : tell every peer "rank r's slab of epoch e has landed", wait until every peer has said so, advance
 *               the counter.
 *     Work enqueued behind it on the same stream sees the whole gathered buffer;
 *   - every wait is BOUNDED (timeout_us of the device's wall clock): a peer that never arrives sets a bit in the
 *     status word instead of hanging the GPU; fp8mi_peer_status reads it.
 *
 * Conventions as in fp8mi.h: return 0 = ok, negative = argument error (the FP8MI_E_* values), positive = a hipError_t;
 * fp8mi_peer_last_error() is the thread-local message.  Unlike the compute library, the SETUP calls here (alloc / free
 * / export / open / close / ctx_create / ctx_destroy / status) allocate or block and are not capturable; the data-path
 * call fp8mi_peer_allgather only enqueues.
 */
#ifndef FP8MI_PEER_H
#define FP8MI_PEER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FP8MI_PEER_VERSION 0x000100
#define FP8MI_PEER_HANDLE_BYTES 64   /* sizeof(hipIpcMemHandle_t) */
#define FP8MI_PEER_MAX_RANKS 16
#define FP8MI_PEER_FLAG_BYTES 4096   /* size of a flag block */

/* status bits (fp8mi_peer_status) */
#define FP8MI_PEER_TIMEOUT_READY 0x1u /* a peer did not become ready within timeout_us: its slab was NOT pushed */
#define FP8MI_PEER_TIMEOUT_DONE 0x2u  /* a peer's slab did not land within timeout_us                           */

typedef struct fp8mi_peer_ctx fp8mi_peer_ctx;

int fp8mi_peer_version(void);
const char *fp8mi_peer_last_error(void);

/* Device memory on the current device, zero-filled, IPC-exportable.  flag_block != 0: uncached memory for a flag
 * block (`bytes` is then ignored: FP8MI_PEER_FLAG_BYTES). */
int fp8mi_peer_alloc(int64_t bytes, int flag_block, void **ptr_out);
int fp8mi_peer_free(void *ptr);

/* IPC: export an allocation of fp8mi_peer_alloc (64 opaque bytes, valid in any process of this host); open / close one
 * exported by ANOTHER process (peer access to the owning device is enabled on first touch). */
int fp8mi_peer_export(void *ptr, void *handle_out);
int fp8mi_peer_open(const void *handle, void **ptr_out);
int fp8mi_peer_close(void *ptr);

/* A context of `world` ranks (2..FP8MI_PEER_MAX_RANKS), this process being `rank`.  data_ptrs[i] / flag_ptrs[i]: rank
 * i's data buffer / flag block AS MAPPED IN THIS PROCESS (own allocation for i == rank, fp8mi_peer_open for the others);
 * host arrays, copied.  data_bytes: size of every data buffer. */
int fp8mi_peer_ctx_create(int world, int rank, void *const *data_ptrs, void *const *flag_ptrs, int64_t data_bytes,
                          fp8mi_peer_ctx **ctx_out);
int fp8mi_peer_ctx_destroy(fp8mi_peer_ctx *ctx);

/* Enqueue on `stream`: this rank's slab [offset, offset + bytes) of its data buffer goes to the same range of every
 * peer's buffer, and the call's end on the stream is the point where every peer's slab of the same call has arrived
 * here.  EVERY rank must make the same sequence of calls (its own offset; the slabs must not overlap).  offset and
 * bytes are multiples of 16.  timeout_us bounds each wait (0 = 30 s). */
int fp8mi_peer_allgather(fp8mi_peer_ctx *ctx, int64_t offset, int64_t bytes, int64_t timeout_us, void *stream);

/* Blocking: synchronise `stream`, then return the OR of the FP8MI_PEER_TIMEOUT_* bits seen so far (and clear them). */
int fp8mi_peer_status(fp8mi_peer_ctx *ctx, void *stream, uint32_t *status_out);

#ifdef __cplusplus
}
#endif
#endif
