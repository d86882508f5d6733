#!/usr/bin/env python3
"""Dispatch check: per-dispatch kernel time of auto vs every forced tile kernel over a grid of shapes.
    python tools/sweep_dispatch.py [bf16|f32]"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fp8-mps-metal_amd")]
import torch, fp8_mi355x_lib as L
out = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dev = torch.device("cuda:0"); lib = L.load()
ws = torch.zeros(int(lib.fp8mi_scaled_mm_workspace_bytes()), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
s1 = torch.full((1,), 0.01, device=dev)
g = torch.Generator(device=dev).manual_seed(1)
SHAPES = [(1024, 1024, 1024), (2048, 2048, 2048), (3072, 3072, 3072), (4096, 4096, 4096), (6144, 6144, 6144),
          (1024, 4096, 4096), (2048, 4096, 4096), (768, 3072, 4096), (1536, 3072, 4096), (4096, 3072, 3072),
          (4096, 3072, 9216), (4096, 12288, 3072), (256, 3072, 12288), (512, 3072, 3072), (1024, 8192, 1024)]
KERNELS = [(L.KERNEL_GEMM_128x64, 1, "128x64"), (L.KERNEL_GEMM_128x64, 0, "128x64+sk"), (L.KERNEL_GEMM_128, 1, "128x128"),
           (L.KERNEL_GEMM_128, 0, "128x128+sk"), (L.KERNEL_GEMM_256, 1, "256x256"), (9, 1, "256x128"), (0, 0, "auto")]
for (M, K, N) in SHAPES:
    A = torch.randint(0, 120, (M, K), dtype=torch.uint8, device=dev, generator=g)
    nb = min(24, max(2, (320 << 20) // (N * K)))
    Bs = [torch.randint(0, 120, (N, K), dtype=torch.uint8, device=dev, generator=g) for _ in range(nb)]
    C = torch.empty(M, N, dtype=torch.float32 if out == "f32" else torch.bfloat16, device=dev)
    res = []
    for kid, split, name in KERNELS:
        def run(i):
            L.check(lib.fp8mi_scaled_mm_ws(A.data_ptr(), Bs[i % nb].data_ptr(), C.data_ptr(), s1.data_ptr(), s1.data_ptr(), None, None,
                                           M, N, K, K, K, N, 0, 0, 0 if out == "f32" else 2, 0, 0, kid, split, ws.data_ptr(), ws.numel(), st), "mm")
        for i in range(nb + 2): run(i)  # touch every weight buffer once (first-touch TLB misses are not the kernel's)
        torch.cuda.synchronize()
        reps = 24
        with L.kernel_timer(reps) as kt:
            for i in range(reps): run(i)
        torch.cuda.synchronize()
        ms = sorted(kt.ms); med = ms[len(ms) // 2] * 1e3
        res.append((name, med))
    best = min(r[1] for r in res[:-1])
    print(f"M={M:5d} K={K:5d} N={N:5d}: " + "  ".join(f"{n} {t:7.1f}" for n, t in res) +
          f"   | auto/best {res[-1][1] / best:.2f}  ({2.0 * M * N * K / (res[-1][1] * 1e-6) / 1e12:.0f} TF)", flush=True)
    del Bs, A, C
    torch.cuda.empty_cache()
