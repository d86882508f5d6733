#!/usr/bin/env python3
"""Random regret sweep: AUTO against every product kernel that accepts the shape, on shapes drawn from LLM / diffusion dimensions.  Prints one line per shape
and the worst AUTO / best ratios at the end (what the dispatch rules are re-fitted from).
    [OUT=f32] python tools/sweep_regret.py [seed] [shapes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fp8-mps-metal_amd")]
import numpy as np, torch, fp8_mi355x_lib as L
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
OUT_F32 = os.environ.get("OUT", "bf16") == "f32"
NOWS = os.environ.get("NOWS", "0") == "1"   # callers without a split-K workspace (a sharded linear's calls, a first call inside a graph capture): every kernel runs unsplit
rng = np.random.default_rng(seed)
order_rng = np.random.default_rng(seed + 100000)
dev = torch.device("cuda:0"); lib = L.load()
ws = torch.zeros(int(lib.fp8mi_scaled_mm_workspace_bytes()), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
s1 = torch.full((1,), 0.01, device=dev)
g = torch.Generator(device=dev).manual_seed(seed)
POOL = torch.randint(0, 120, (1 << 30,), dtype=torch.uint8, device=dev, generator=g)   # weight bytes: every shape's rotating buffers are slices of this
DIMS = [1024, 1536, 2048, 2560, 3072, 4096, 5120, 6144, 7168, 8192, 9216, 10240, 12288, 13824, 14336, 16384, 28672]
MS = [1, 2, 4, 8, 12, 16, 24, 32, 48, 64, 96, 128, 160, 192, 256, 320, 384, 512, 640, 768, 1024, 1280, 1536, 2048, 3072, 4096, 8192]
if os.environ.get("MS"):
    MS = [int(v) for v in os.environ["MS"].split(",")]
if os.environ.get("DIMS", "") == "ext":   # round 4: also small and ragged sizes (the cost model must not extrapolate blindly below K, N = 1024 or off the tile grid)
    DIMS = [128, 256, 384, 512, 640, 768, 1008, 1024, 1280, 1536, 2000, 2048, 2560, 3008, 3072, 3584, 4096, 5008, 5120, 6144, 7168, 8192, 9216, 11008, 12288, 14336]   # (multiples of 16: K % 16 != 0 is the generic kernel's)
    if not os.environ.get("MS"):
        MS = [2, 3, 5, 6, 8, 10, 16, 20, 28, 32, 40, 56, 64, 72, 100, 128, 144, 200, 256, 300, 400, 512, 600, 1000, 1024, 1500, 2048, 3000, 4096, 6000]
NAMES = {L.KERNEL_GEMV: "gemv", L.KERNEL_GEMV_MX: "mx", L.KERNEL_SKINNY: "skinny", L.KERNEL_GEMM_32x32: "32x32", L.KERNEL_GEMM_32x64: "32x64",
         L.KERNEL_GEMM_64x64: "64x64", L.KERNEL_GEMM_64x128: "64x128", L.KERNEL_GEMM_128x64: "128x64", L.KERNEL_GEMM_128: "128", L.KERNEL_GEMM_128D: "128D",
         L.KERNEL_GEMM_256W: "256W", L.KERNEL_GEMM_256x128W: "256x128W"}


def candidates(M, K, N):
    ks = []
    if M == 1: ks.append(L.KERNEL_GEMV)
    if 2 <= M <= 8: ks.append(L.KERNEL_GEMV_MX)
    if 2 <= M <= 64: ks.append(L.KERNEL_SKINNY)
    if M <= 512: ks += [L.KERNEL_GEMM_32x32, L.KERNEL_GEMM_32x64]          # (round 4: the small tiles up to M = 512 / 1024 - one round of 64x64 tiles against a shallow K beats
    if M <= 1024: ks += [L.KERNEL_GEMM_64x64, L.KERNEL_GEMM_64x128]         #  half a round of 128x64: M=256 K=N=4096 9.9 against 13.4 us)
    if M > 1: ks += [L.KERNEL_GEMM_128x64, L.KERNEL_GEMM_128]
    if M > 64 and ((M + 127) // 128) * ((N + 127) // 128) <= 640: ks.append(L.KERNEL_GEMM_128D)
    if M > 128: ks += [L.KERNEL_GEMM_256x128W]
    if M > 256: ks += [L.KERNEL_GEMM_256W]
    return ks


rows = []
done = 0
FIXED = [tuple(int(v) for v in sh.split(",")) for sh in os.environ.get("SHAPES", "").split(";") if sh]   # SHAPES="M,K,N;M,K,N": these shapes instead of random ones
if FIXED:
    count = len(FIXED)
while done < count:
    M, K, N = FIXED[done] if FIXED else (int(rng.choice(MS)), int(rng.choice(DIMS)), int(rng.choice(DIMS)))
    if not FIXED and (2.0 * M * N * K > 2.5e12 or N * K > (512 << 20)): continue
    done += 1
    # COLD weights for every shape (round 4): the weight buffers are slices of one 1 GiB pool, at least 320 MiB of them (more than the 256 MiB Infinity
    # Cache) in rotation.  Until round 4 the count was capped at 16 buffers, so every matrix below 16 MiB - a third of the standard shapes, three
    # quarters of the small-dimension ones - was timed with its weights resident in the Infinity Cache, which no model with more than 256 MiB of
    # weights ever sees (and which bench.py never did); on config C3 that flipped the ranking of 128x64 against 64x128 tiles.
    nb = max(2, min(640, -(-(320 << 20) // (N * K))))
    nb = min(nb, POOL.numel() // (N * K))
    Bs = [POOL[i * N * K:(i + 1) * N * K].view(N, K) for i in range(nb)]
    A = torch.randint(0, 120, (M, K), dtype=torch.uint8, device=dev, generator=g)
    C = torch.empty(M, N, dtype=torch.float32 if OUT_F32 else torch.bfloat16, device=dev)
    res = {}
    picked = lib.fp8mi_choose_kernel(M, N, K, K, K, N, 0 if OUT_F32 else 2, 0 if NOWS else 1, 0)
    cursor = [0]   # every launch takes the NEXT buffer of the rotation (warm-ups and timed ones alike): a buffer comes round again only after >= 320 MiB of others
    cands = candidates(M, K, N)
    order_rng.shuffle(cands)   # (a kernel's position in the sequence is not always the same: what ran before it moves the clocks)
    for kid in [0] + cands + [0]:
        def run():
            i = cursor[0] % nb
            cursor[0] += 1
            return lib.fp8mi_scaled_mm_ws(A.data_ptr(), Bs[i].data_ptr(), C.data_ptr(), s1.data_ptr(), s1.data_ptr(), None, None,
                                          M, N, K, K, K, N, 0, 0, 0 if OUT_F32 else 2, 0, 0, kid, 1 if NOWS else 0, None if NOWS else ws.data_ptr(), 0 if NOWS else ws.numel(), st)
        if run() != 0: continue
        for _ in range(10): run()
        torch.cuda.synchronize()
        reps = 16
        with L.kernel_timer(reps) as kt:
            for _ in range(reps): run()
        torch.cuda.synchronize()
        ms = sorted(kt.ms); t = ms[len(ms) // 2] * 1e3
        res[kid] = min(res.get(kid, 1e30), t)   # AUTO is timed first and last: the better of the two
    if len(res) < 2:   # no forced kernel takes the shape
        continue
    best_k, best_t = min(((k, t) for k, t in res.items() if k != 0), key=lambda kv: kv[1])
    ratio = res[0] / best_t
    rows.append((ratio, M, K, N, picked, best_k, res[0], best_t))
    print(f"M={M:5d} K={K:5d} N={N:5d}: auto({NAMES.get(picked, picked)}) {res[0]:7.1f}  " + "  ".join(f"{NAMES[k]} {t:.1f}" for k, t in res.items() if k != 0) +
          f"   | auto/best {ratio:.2f}", flush=True)
    del Bs, A, C
rows.sort(reverse=True)
print("# worst 25:")
for r, M, K, N, pk, bk, ta, tb in rows[:25]:
    print(f"#  {r:.2f}  M={M} K={K} N={N}: auto = {NAMES.get(pk, pk)} {ta:.1f} us, best = {NAMES[bk]} {tb:.1f} us")
import statistics
print(f"# {len(rows)} shapes: median regret {statistics.median(r[0] for r in rows):.3f}, > 1.10: {sum(r[0] > 1.10 for r in rows)}, > 1.20: {sum(r[0] > 1.20 for r in rows)}")
