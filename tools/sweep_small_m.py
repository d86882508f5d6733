#!/usr/bin/env python3
"""Small-batch dispatch check: auto vs skinny vs split-K tile kernels.   python tools/sweep_small_m.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fp8-mps-metal_amd")]
import torch, fp8_mi355x_lib as L
dev = torch.device("cuda:0"); lib = L.load()
ws = torch.zeros(int(lib.fp8mi_scaled_mm_workspace_bytes()), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
s1 = torch.full((1,), 0.01, device=dev)
g = torch.Generator(device=dev).manual_seed(1)
KN = [(4096, 4096), (14336, 4096), (4096, 14336), (8192, 8192), (3072, 12288), (12288, 3072), (2048, 2048)]
MS = [2, 4, 8, 16, 32, 48, 64, 96, 128]
KERNELS = [(L.KERNEL_SKINNY, 1, "skinny"), (L.KERNEL_GEMM_64x128, 0, "64x128+sk"), (L.KERNEL_GEMM_128x64, 0, "128x64+sk"), (0, 0, "auto")]
for (K, N) in KN:
    nb = min(24, max(2, (320 << 20) // (N * K)))
    Bs = [torch.randint(0, 120, (N, K), dtype=torch.uint8, device=dev, generator=g) for _ in range(nb)]
    for M in MS:
        A = torch.randint(0, 120, (M, K), dtype=torch.uint8, device=dev, generator=g)
        C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        res = []
        for kid, split, name in KERNELS:
            if kid == L.KERNEL_SKINNY and M > 64:
                res.append((name, float("nan"))); continue
            def run(i):
                L.check(lib.fp8mi_scaled_mm_ws(A.data_ptr(), Bs[i % nb].data_ptr(), C.data_ptr(), s1.data_ptr(), s1.data_ptr(), None, None,
                                               M, N, K, K, K, N, 0, 0, 2, 0, 0, kid, split, ws.data_ptr(), ws.numel(), st), "mm")
            for i in range(nb + 2): run(i)
            torch.cuda.synchronize()
            reps = 24
            with L.kernel_timer(reps) as kt:
                for i in range(reps): run(i)
            torch.cuda.synchronize()
            ms = sorted(kt.ms); res.append((name, ms[len(ms) // 2] * 1e3))
        best = min(t for _, t in res[:-1] if t == t)
        print(f"K={K:5d} N={N:5d} M={M:3d}: " + "  ".join(f"{n} {t:6.1f}" for n, t in res) + f"   | auto/best {res[-1][1] / best:.2f}", flush=True)
    del Bs
    torch.cuda.empty_cache()
