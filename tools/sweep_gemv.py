#!/usr/bin/env python3
"""M = 1: AUTO vs forced vec-mat launch shapes over (K, N).  FP8MI_LIB_PATH=.../libfp8mi_diag.so python tools/sweep_gemv.py [ids ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fp8-mps-metal_amd")]
import torch, fp8_mi355x_lib as L
dev = torch.device("cuda:0"); lib = L.load()
st = torch.cuda.current_stream().cuda_stream
s1 = torch.full((1,), 0.01, device=dev)
g = torch.Generator(device=dev).manual_seed(1)
ids = [int(x) for x in sys.argv[1:]] or [0]
for (K, N) in [(14336, 4096), (14336, 14336), (16384, 4096), (12288, 3072), (11008, 4096), (8192, 8192), (8192, 28672), (7168, 7168), (5120, 5120), (4096, 4096), (4096, 14336), (4096, 11008), (3072, 12288)]:
    nb = min(24, max(2, (320 << 20) // (N * K)))
    Bs = [torch.randint(0, 120, (N, K), dtype=torch.uint8, device=dev, generator=g) for _ in range(nb)]
    A = torch.randint(0, 120, (1, K), dtype=torch.uint8, device=dev, generator=g)
    C = torch.empty(1, N, dtype=torch.float32, device=dev)
    res = []
    for kid in ids:
        def run(i):
            return lib.fp8mi_scaled_mm_ws(A.data_ptr(), Bs[i % nb].data_ptr(), C.data_ptr(), s1.data_ptr(), s1.data_ptr(), None, None,
                                          1, N, K, K, K, N, 0, 0, 0, 0, 0, kid, 1, None, 0, st)
        if run(0) != 0:
            res.append(float("nan")); continue
        for i in range(nb + 4): run(i)
        torch.cuda.synchronize()
        with L.kernel_timer(48) as kt:
            for i in range(48): run(i)
        torch.cuda.synchronize()
        ms = sorted(kt.ms); res.append(ms[len(ms) // 2] * 1e3)
    best = min(t for t in res if t == t)
    print(f"K={K:5d} N={N:5d}: " + "  ".join(f"{k}:{t:6.2f}" for k, t in zip(ids, res)) + f"   | first/best {res[0] / best:.3f}   best {(N * K + K + 4 * N) / best / 1e6:5.2f} TB/s", flush=True)
    del Bs
    torch.cuda.empty_cache()
