#!/usr/bin/env python3
"""AUTO vs forced tile kernels with the automatic K split over (K, N) x M grids (default: the decode regime, 9 <= M <= 128 against deep K).
    [FP8MI_LIB_PATH=.../libfp8mi_diag.so] [MS=512,1024] [KN=4096x4096,3072x12288] [OUT=f32] python tools/sweep_decode.py [kernel ids ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fp8-mps-metal_amd")]
import torch, fp8_mi355x_lib as L
dev = torch.device("cuda:0"); lib = L.load()
ws = torch.zeros(int(lib.fp8mi_scaled_mm_workspace_bytes()), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
s1 = torch.full((1,), 0.01, device=dev)
g = torch.Generator(device=dev).manual_seed(1)
KN = [(14336, 4096), (8192, 8192), (4096, 14336), (12288, 3072), (4096, 4096), (7168, 7168)]
if os.environ.get("KN"):   # e.g. KN=4096x4096,3072x12288 (K x N)
    KN = [tuple(int(v) for v in kn.split("x")) for kn in os.environ["KN"].split(",")]
MS = [int(x) for x in os.environ.get("MS", "9,16,32,48,64,96,128").split(",")]
OUT_F32 = os.environ.get("OUT", "bf16") == "f32"
ids = [int(x) for x in sys.argv[1:]] or [0, L.KERNEL_GEMM_64x128, L.KERNEL_GEMM_128x64]
for (K, N) in KN:
    nb = min(24, max(2, (320 << 20) // (N * K)))
    Bs = [torch.randint(0, 120, (N, K), dtype=torch.uint8, device=dev, generator=g) for _ in range(nb)]
    for M in MS:
        A = torch.randint(0, 120, (M, K), dtype=torch.uint8, device=dev, generator=g)
        C = torch.empty(M, N, dtype=torch.float32 if OUT_F32 else torch.bfloat16, device=dev)
        res = []
        for kid in ids:
            def run(i):
                return lib.fp8mi_scaled_mm_ws(A.data_ptr(), Bs[i % nb].data_ptr(), C.data_ptr(), s1.data_ptr(), s1.data_ptr(), None, None,
                                              M, N, K, K, K, N, 0, 0, 0 if OUT_F32 else 2, 0, 0, kid, 0, ws.data_ptr(), ws.numel(), st)
            if run(0) != 0:
                res.append(float("nan")); continue
            for i in range(nb + 2): run(i)
            torch.cuda.synchronize()
            with L.kernel_timer(24) as kt:
                for i in range(24): run(i)
            torch.cuda.synchronize()
            ms = sorted(kt.ms); res.append(ms[len(ms) // 2] * 1e3)
        best = min(t for t in res if t == t)
        print(f"K={K:5d} N={N:5d} M={M:3d}: " + "  ".join(f"{k}:{t:6.1f}" for k, t in zip(ids, res)) + f"   | first/best {res[0] / best:.2f}", flush=True)
    del Bs
    torch.cuda.empty_cache()
