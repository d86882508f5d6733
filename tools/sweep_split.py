#!/usr/bin/env python3
"""Is the launcher's automatic K split (resolve_split, fp8mi_gemm_epi.h: fill the chip when tiles <= CUs / 2, >= 4 ring stages per slice, <= 16) the right count?
Every split-capable tile kernel with forced split_k = 1, 2, 3, 4, 6, 8, 12, 16 against its automatic split and against AUTO, on random small-batch / narrow shapes.
    python tools/sweep_split.py [seed] [shapes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fp8-mps-metal_amd")]
import numpy as np, torch, fp8_mi355x_lib as L
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(seed)
dev = torch.device("cuda:0"); lib = L.load()
ws = torch.zeros(int(lib.fp8mi_scaled_mm_workspace_bytes()), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
s1 = torch.full((1,), 0.01, device=dev)
g = torch.Generator(device=dev).manual_seed(seed)
DIMS = [1024, 1536, 2048, 2560, 3072, 4096, 5120, 6144, 7168, 8192, 10240, 12288, 14336, 16384, 28672]
MS = [8, 16, 32, 48, 64, 96, 128, 160, 192, 256, 320, 384, 512, 768, 1024]
KERNELS = {"32x32": L.KERNEL_GEMM_32x32, "32x64": L.KERNEL_GEMM_32x64, "64x64": L.KERNEL_GEMM_64x64, "64x128": L.KERNEL_GEMM_64x128, "128x64": L.KERNEL_GEMM_128x64, "128D": L.KERNEL_GEMM_128D}
SPLITS = [0, 1, 2, 3, 4, 6, 8, 12, 16]
done = 0
gains = []
while done < count:
    M, K, N = int(rng.choice(MS)), int(rng.choice(DIMS)), int(rng.choice(DIMS))
    if M * N > (1 << 21) or N * K > (256 << 20): continue     # the regime where a split can matter: at most ~2 rounds of 128x64 tiles
    done += 1
    nb = min(16, max(2, (300 << 20) // (N * K)))
    Bs = [torch.randint(0, 120, (N, K), dtype=torch.uint8, device=dev, generator=g) for _ in range(nb)]
    A = torch.randint(0, 120, (M, K), dtype=torch.uint8, device=dev, generator=g)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    def timed(kid, split):
        def run(i):
            return lib.fp8mi_scaled_mm_ws(A.data_ptr(), Bs[i % nb].data_ptr(), C.data_ptr(), s1.data_ptr(), s1.data_ptr(), None, None, M, N, K, K, K, N, 0, 0, 2, 0, 0, kid, split, ws.data_ptr(), ws.numel(), st)
        if run(0) != 0: return None
        for i in range(nb + 2): run(i)
        torch.cuda.synchronize()
        with L.kernel_timer(12) as kt:
            for i in range(12): run(i)
        torch.cuda.synchronize()
        ms = sorted(kt.ms); return ms[len(ms) // 2] * 1e3
    t_auto = timed(0, 0)
    res = {}
    for name, kid in KERNELS.items():
        if name in ("32x32", "32x64") and M > 512: continue
        if name == "128D" and M <= 64: continue
        for s in SPLITS:
            t = timed(kid, s)
            if t is not None: res[(name, s)] = t
    best_auto = min((t, k) for k, t in res.items() if k[1] == 0)
    best_any = min((t, k) for k, t in res.items())
    gains.append(best_auto[0] / best_any[0])
    print(f"M={M:5d} K={K:5d} N={N:5d}: AUTO {t_auto:6.1f} | best with the automatic split {best_auto[1][0]} {best_auto[0]:6.1f} | best with any split {best_any[1][0]} x{best_any[1][1]} {best_any[0]:6.1f}  (x{best_auto[0] / best_any[0]:.2f})  " +
          "  ".join(f"{n}:" + ",".join(f"{res.get((n, s), 0):.0f}" for s in SPLITS) for n in KERNELS if (n, 0) in res), flush=True)
    del Bs, A, C
    torch.cuda.empty_cache()
import statistics
print(f"# {len(gains)} shapes: best(any forced split) beats best(automatic split) by median x{statistics.median(gains):.3f}; > 1.05: {sum(x > 1.05 for x in gains)}, > 1.10: {sum(x > 1.10 for x in gains)}, > 1.20: {sum(x > 1.20 for x in gains)}")
