/*
 * fp8mi - MI355X (gfx950) FP8 e4m3fn scaled-matmul and cast kernels, C ABI.
 *
 * This is the drop-in boundary of the build: a shared library
 * (fp8-mps-metal_amd/libfp8mi.so) with plain-C entry points - device pointers,
 * sizes and a HIP stream; no torch or C++ types - that a host binds with
 * ctypes / cgo / JNI.  It is the MI355X-native counterpart of the reference's
 * pybind11 module `fp8_metal` (fp8_bridge.cpp:361-371) and of the four kernel
 * launch sites of fp8_mps_native.py, but pointer-level: nothing is staged
 * through the CPU, nothing is allocated, nothing synchronises.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer on the current HIP device unless said
 *     otherwise; `stream` is a hipStream_t passed as void* (NULL = default);
 *   - every call only ENQUEUES work on `stream` and returns; it never blocks,
 *     never allocates, and is safe to capture into a HIP graph;
 *   - return value 0 = ok; negative = argument error (FP8MI_E_*); positive =
 *     a hipError_t from the launch.  fp8mi_last_error() returns a
 *     thread-local, human-readable message for the last non-zero return;
 *   - the compute entry points keep no state (no globals besides that message
 *     and a per-device cache of the CU count), so calls are re-entrant from any
 *     thread.  The two measurement hooks at the end of this header
 *     (fp8mi_profile_begin / _end) DO keep per-thread state - an event pool and
 *     an "a profile is open" flag that every launch on that thread consults;
 *     they exist for bench.py and are not meant for production call paths;
 *   - all element counts are 64-bit (the reference's `uint count`,
 *     fp8_matmul.metal:218,231, caps at 2^32-1).
 *
 * Semantics are the REFERENCE's (audiohacking/fp8-mps-metal), not OCP/torch's,
 * wherever they differ; the non-default modes give the OCP behaviour:
 *   decode: NaN bytes 0x7F/0xFF -> +0.0              (fp8_matmul.metal:21)
 *   encode: clamp-not-carry, flush below 2^-9, saturate to 0x7E, -0.0 -> 0x00
 *                                                    (fp8_matmul.metal:44-92)
 */
#ifndef FP8MI_H
#define FP8MI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FP8MI_VERSION 0x000400 /* 0.4.0: + fp8mi_predict_kernel_us (the dispatch is one cost model); 0.3.0: + fp8mi_workspace_reset, FP8MI_EPILOGUE_TRANSPOSED, kernel ids GEMV_FP32 / GEMV_MX / GEMM_256W */

/* element types of non-fp8 operands */
enum { FP8MI_F32 = 0, FP8MI_F16 = 1, FP8MI_BF16 = 2 };

/* OR into `bias_dtype`: the caller computes the TRANSPOSED product C^T = B . A^T (what a
 * rank of an N-column-sharded linear does, so that its block of the output is contiguous:
 * the weight shard is passed as `A`, the activations as `B_nk`).  The epilogue then
 * multiplies by scale_b first and scale_a second and takes bias[m] (M elements, per
 * output row) - bit for bit what the untransposed fused epilogue
 * ((acc * s_activation) * s_weight + bias[weight row]) would have stored. */
#define FP8MI_EPILOGUE_TRANSPOSED 0x100

/* scale layouts */
enum { FP8MI_SCALE_TENSOR = 0, /* one float                               */
       FP8MI_SCALE_ROW = 1 };  /* one float per row of A (M) / of B (N)   */

/* what a NaN byte (0x7F / 0xFF) in A or B means */
enum { FP8MI_NAN_ZERO = 0,       /* reference: decodes to +0.0 (fp8_matmul.metal:21)   */
       FP8MI_NAN_PROPAGATE = 1 };/* OCP / torch: NaN, poisons its dot products        */

/* float -> fp8 rounding/saturation rules */
enum { FP8MI_ENC_REFERENCE = 0,  /* fp8_matmul.metal:44-92 (see header comment)        */
       FP8MI_ENC_RNE = 1 };      /* OCP round-to-nearest-even, overflow -> NaN (torch) */

/* kernel selection for fp8mi_scaled_mm_ex (testing / benchmarking) */
enum { FP8MI_KERNEL_AUTO = 0,
       FP8MI_KERNEL_GEMV = 1,      /* M == 1 wavefront-reduced vec-mat (fp32 FMA for K <= 4096, matrix core above) */
       FP8MI_KERNEL_GEMM_128 = 2,  /* 128x128x128 LDS-tiled fp8 MFMA (mid sizes)      */
       FP8MI_KERNEL_GENERIC = 3,   /* any shape / alignment, one wave per output      */
       FP8MI_KERNEL_GEMM_256 = 4,  /* 256x256x128 LDS-tiled fp8 MFMA (large M,N)      */
       FP8MI_KERNEL_GEMM_128x64 = 5, /* 128x64x128 tile (few tiles: one per CU)       */
       FP8MI_KERNEL_SKINNY = 6,    /* 1 <= M <= 64 weight-streaming MFMA              */
       FP8MI_KERNEL_GEMM_64x128 = 14, /* 64x128x128 tile (M <= 64, deep K)            */
       FP8MI_KERNEL_GEMV_FP32 = 18,  /* M == 1, IEEE fp32 accumulation at every K     */
       FP8MI_KERNEL_GEMV_MX = 19,    /* 2 <= M <= 8 on the vec-mat's weight-streaming structure */
       FP8MI_KERNEL_GEMM_256W = 20, /* 256x256 tile, one wave per SIMD, hand-scheduled K loop: any M, N (a multiple of 4 fp32 / 8 half columns), K >= 256 (K % 16 == 0; a partial last K-step since round 3) */
       FP8MI_KERNEL_GEMM_256x128W = 21, /* the same on 256x128 tiles (shapes that give 256x256 tiles less than a round) */
       FP8MI_KERNEL_GEMM_64x64 = 22,  /* 64x64x128 tile, 8 waves (33 <= M <= 64 against deep K, and up to M = 128 while the tile grid is small; with split-K) */
       FP8MI_KERNEL_GEMM_32x64 = 23,  /* 32x64x128 tile, 4 waves (9 <= M <= 32: the decode regime; with split-K) */
       FP8MI_KERNEL_GEMM_32x32 = 24,  /* 32x32x128 tile, 2 waves (M <= 32 against K, N <= 8192: N / 32 tiles fill the chip with fewer K slices) */
       FP8MI_KERNEL_GEMM_128D = 25 }; /* 128x128x128 tile on a DEEP ring (4 x 32 KiB, one workgroup per CU): shapes that give at most one 128x128 tile per CU */
/* Other ids exist only in the diagnostic build of the library (libfp8mi_diag.so:
 * schedule variants, the producer/consumer kernel and its ablations, kept for
 * A/B timing - tools/README.md); the product library rejects them. */

/* error codes (negative returns) */
enum { FP8MI_OK = 0,
       FP8MI_E_NULL = -1,     /* required pointer is NULL                  */
       FP8MI_E_SHAPE = -2,    /* negative size or leading dim too small    */
       FP8MI_E_ENUM = -3,     /* unknown dtype / mode / kernel             */
       FP8MI_E_UNSUPPORTED = -4 /* forced kernel cannot run this problem   */ };

/*
 * C[m,n] = cast( ((sum_k dec(A[m,k]) * dec(B[n,k])) * sa * sb + bias[n]) * scale_result )
 *
 * Replaces: fp8_scaled_matmul_kernel (fp8_matmul.metal:99-147),
 *           fp8_scaled_vecmat_kernel (fp8_matmul.metal:155-210), their launch
 *           sites fp8_mps_native.py:41-95 / fp8_bridge.cpp:165-259, the kernel
 *           choice fp8_scaled_mm_auto (fp8_mps_native.py:193-210) and the three
 *           elementwise passes of the epilogue in fp8_mps_patch.py:94-104
 *           (fused here, same order: + bias, * scale_result, cast).
 *
 * A        (M,K) fp8 e4m3fn bytes, row-major, leading dimension lda >= K
 * B_nk     (N,K) fp8 e4m3fn bytes, row-major (i.e. torch's column-major
 *          `other` (K,N) seen through .t()), leading dimension ldb >= K
 * C        (M,N) out_dtype, row-major, leading dimension ldc >= N (elements)
 * scale_a  float[1] (FP8MI_SCALE_TENSOR) or float[M] (FP8MI_SCALE_ROW)
 * scale_b  float[1] or float[N]
 * bias     NULL or [N] of bias_dtype ([M] with FP8MI_EPILOGUE_TRANSPOSED);  scale_result NULL or float[1]
 * Accumulators are float32.  On the matrix-core kernels (M > 1; M == 1 with K > 4096) the
 * sum inside an instruction is the gfx950 fp8 MFMA's: exact products, addends more than
 * ~2^13 below the largest of their group of 8 truncated (|err| <= 1e-3 sum|a||b| worst
 * case, ~2e-5 rms on random data, exact on operands within a 2^12 product range) - two
 * orders of magnitude inside the reference's own 4 % gate.  FP8MI_KERNEL_GEMV_FP32 and
 * FP8MI_KERNEL_GENERIC accumulate in IEEE fp32 like the reference's kernels
 * (fp8_matmul.metal:116-141, 177-199).  M == 0 or N == 0 is a no-op; K == 0 writes the
 * epilogue of a zero sum.
 */
int fp8mi_scaled_mm(const uint8_t *A, const uint8_t *B_nk, void *C,
                    const float *scale_a, const float *scale_b,
                    const void *bias, const float *scale_result,
                    int64_t M, int64_t N, int64_t K,
                    int64_t lda, int64_t ldb, int64_t ldc,
                    int scale_a_mode, int scale_b_mode,
                    int out_dtype, int bias_dtype, int nan_mode,
                    void *stream);

/* Same, with the kernel forced (FP8MI_KERNEL_*); FP8MI_E_UNSUPPORTED if the
 * forced kernel cannot run the problem (e.g. GEMV with M != 1). */
int fp8mi_scaled_mm_ex(const uint8_t *A, const uint8_t *B_nk, void *C,
                       const float *scale_a, const float *scale_b,
                       const void *bias, const float *scale_result,
                       int64_t M, int64_t N, int64_t K,
                       int64_t lda, int64_t ldb, int64_t ldc,
                       int scale_a_mode, int scale_b_mode,
                       int out_dtype, int bias_dtype, int nan_mode,
                       int kernel, void *stream);

/*
 * Split-K.  When M x N yields far fewer output tiles than the device has CUs (small M,
 * deep K: the decode / small-batch regime) the tile kernels can cut K into
 * `split_k` slices, one workgroup per (tile, slice); the slices' fp32 partial
 * tiles meet in `workspace`, and the last workgroup of a tile to finish adds
 * them in slice order (run-to-run reproducible) and runs the fused epilogue.
 * No counterpart in the reference (its kernels are one thread per output,
 * fp8_matmul.metal:99-147).
 *
 * workspace        device buffer, 16-byte aligned, used by ONE launch at a time
 *                  (launches on one stream may share it; concurrent streams
 *                  need their own).  Its first FP8MI_WS_COUNTER_BYTES bytes
 *                  (the tiles' arrival counters) must be zero before the first
 *                  launch that uses it - fp8mi_workspace_reset() - and every
 *                  launch that completes leaves them zero.  A launch that is
 *                  ABORTED mid-flight (device fault) can leave a counter
 *                  non-zero, and later launches on that workspace would then
 *                  mis-reduce silently: reset the workspace as part of any
 *                  error recovery.  NULL: never split.
 * workspace_bytes  its size; fp8mi_scaled_mm_workspace_bytes() is enough for
 *                  every problem the library would split on its own.  A
 *                  workspace that is too small silently disables the split.
 * split_k          0: library decides; 1: no split; > 1: that many slices
 *                  (clamped to the largest count K and the workspace allow).
 * fp8mi_scaled_mm / _ex are this call with workspace == NULL.
 */
#define FP8MI_WS_COUNTER_BYTES 4096
int64_t fp8mi_scaled_mm_workspace_bytes(void);
/* Enqueue a memset of the counter block (FP8MI_WS_COUNTER_BYTES at the head of
 * `workspace`) on `stream`: once after allocating a workspace, and after any
 * aborted launch.  Graph-capturable (a memset node). */
int fp8mi_workspace_reset(void *workspace, int64_t workspace_bytes, void *stream);
int fp8mi_scaled_mm_ws(const uint8_t *A, const uint8_t *B_nk, void *C,
                       const float *scale_a, const float *scale_b,
                       const void *bias, const float *scale_result,
                       int64_t M, int64_t N, int64_t K,
                       int64_t lda, int64_t ldb, int64_t ldc,
                       int scale_a_mode, int scale_b_mode,
                       int out_dtype, int bias_dtype, int nan_mode,
                       int kernel, int split_k,
                       void *workspace, int64_t workspace_bytes,
                       void *stream);


/*
 * out[i] = cast( half(dec(in[i])) * half(scale) )      (scale NULL = no multiply)
 *
 * Replaces: fp8_to_half_kernel (fp8_matmul.metal:215-223) + the separate
 *           `output * scale.to(float16)` pass of fp8_dequantize
 *           (fp8_mps_native.py:98-124; fp8_bridge.cpp:265-306).  The product
 *           is formed in float16 exactly as there; out_dtype F32/BF16 then
 *           widens/rounds that half (the reference's `.to(dtype)`,
 *           fp8_mps_patch.py:219-221).
 */
int fp8mi_dequant(const uint8_t *in, void *out, const float *scale,
                  int64_t count, int out_dtype, void *stream);

/* The same entry point under the name SURVEY.md 8(b) item 2 gives it (the half output of fp8_to_half_kernel is the
 * default use; out_dtype still selects F16 / F32 / BF16).  Two deviations from 8(b) as written, both on purpose:
 * the symbol was shortened to fp8mi_dequant (this alias keeps the contract's name linkable), and every leading
 * dimension / count in this header is int64_t where 8(b) says `int` (a (M,K) slab of a > 2 GiB buffer needs it). */
int fp8mi_dequant_f16(const uint8_t *in, void *out, const float *scale_or_null,
                      int64_t count, int out_dtype, void *stream);

/*
 * out[i] = enc( float32(in[i]) * prescale )            (prescale NULL = no multiply)
 *
 * Replaces: float_to_fp8_kernel (fp8_matmul.metal:228-236) and its launch in
 *           fp8_encode (fp8_mps_native.py:127-155; value-preserving, used by
 *           .to()/.copy_(), fp8_mps_patch.py:191,283); with `prescale` also
 *           the `inp * scale` pass of fp8_quantize (fp8_mps_native.py:179).
 *           f16/bf16 inputs are widened in-kernel (the reference converts to
 *           float32 first, fp8_mps_native.py:142).
 */
int fp8mi_encode(const void *in, int in_dtype, uint8_t *out, const float *prescale,
                 int64_t count, int encode_mode, void *stream);

/* *out = max_i |in[i]| as float32 (0 for count == 0; NaNs are ignored).
 * Replaces: `inp.abs().max().item()` of fp8_quantize (fp8_mps_native.py:174)
 * without the host read-back. */
int fp8mi_amax(const void *in, int in_dtype, float *out, int64_t count, void *stream);

/*
 * Amax-scaled quantisation, entirely on the device (no host sync, two kernels):
 *   amax = max|in|; scale = amax > 0 ? 448/amax : 1   (evaluated in double)
 *   out[i] = enc(float32(in[i]) * float32(scale));  scales[0] = amax,
 *   scales[1] = float32(1/scale)  (the inverse scale _scaled_mm consumes)
 * Replaces: fp8_quantize (fp8_mps_native.py:158-190; fp8_bridge.cpp:312-356).
 * `scales` is float[2] device memory owned by the caller.
 */
int fp8mi_quantize(const void *in, int in_dtype, uint8_t *out, float *scales,
                   int64_t count, int encode_mode, void *stream);

typedef struct fp8mi_device_info {
    int compute_units;       /* multiProcessorCount                       */
    int clock_khz;           /* max engine clock                          */
    int memory_clock_khz;
    int memory_bus_bits;
    int l2_bytes;
    int lds_bytes_per_cu;    /* maxSharedMemoryPerMultiProcessor          */
    int wavefront_size;
    int64_t total_memory;
    char arch[64];           /* gcnArchName, e.g. "gfx950:sramecc+:xnack-" */
    char name[128];
} fp8mi_device_info_t;

/* Host-side query of HIP device `device` (for the roofline harness). */
int fp8mi_device_info(int device, fp8mi_device_info_t *out);

/*
 * Per-dispatch kernel timing for the measurement harness (bench.py).  Between
 * fp8mi_profile_begin(n) and fp8mi_profile_end() every kernel this library
 * launches FROM THE CALLING THREAD carries its own start/stop HIP event pair
 * filled from the dispatch packet's timestamps (hipExtLaunchKernelGGL) - the
 * clock rocprofv3 --kernel-trace reads - up to n launches.  _end() waits for
 * the last profiled launch, writes up to `cap` durations (milliseconds, launch
 * order) to the HOST array ms_out and returns how many launches were timed (or
 * a negative error).  Not for use inside graph capture.  No reference
 * counterpart: the reference times with time.perf_counter() around
 * torch.mps.synchronize() (test_fp8_metal.py:248-255).
 */
int fp8mi_profile_begin(int max_launches);
int fp8mi_profile_end(float *ms_out, int cap);

int fp8mi_version(void);
const char *fp8mi_last_error(void);

/* Which kernel FP8MI_KERNEL_AUTO runs for a problem of this shape (host-only; pointers are assumed 16-byte aligned):
 * returns an FP8MI_KERNEL_* id, or a negative error for an invalid argument.  For tests and for callers that want to log
 * the dispatch; no counterpart in the reference (its choice is `M == 1` in fp8_mps_native.py:193-210). */
int fp8mi_choose_kernel(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int out_dtype,
                        int has_workspace, int split_k);

/* The dispatch's cost model itself (round 4; host-only): the time in microseconds it predicts for `kernel` on this problem on a device of
 * `compute_units` CUs (0 = the current device's), or a negative value when the kernel does not take the problem or is not offered for it
 * (FP8MI_KERNEL_GEMV / _GENERIC / _AUTO are not priced: M = 1 is a rule, the generic kernel the last resort).  fp8mi_choose_kernel returns
 * the id with the smallest prediction.  For tests (tests/golden/dispatch_times_cold_*.json) and for callers that log the dispatch. */
double fp8mi_predict_kernel_us(int kernel, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int out_dtype,
                               int has_workspace, int split_k, int compute_units);

#ifdef __cplusplus
}
#endif
#endif /* FP8MI_H */
