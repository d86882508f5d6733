#!/usr/bin/env python3
"""Does a power-of-two row stride cost time?  AUTO on (M, K, N) with the operands' row stride K (as torch allocates them) against K + PAD bytes
(the C ABI takes lda / ldb): median per-dispatch us of both and their ratio.
    [PADS=256,2048] [MS=1,16,64,128,512,2048] [KN=8192x8192,...] [KID=0] python tools/sweep_pad.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fp8-mps-metal_amd")]
import torch, fp8_mi355x_lib as L
dev = torch.device("cuda:0"); lib = L.load()
ws = torch.zeros(int(lib.fp8mi_scaled_mm_workspace_bytes()), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
s1 = torch.full((1,), 0.01, device=dev)
g = torch.Generator(device=dev).manual_seed(1)
KN = [tuple(int(v) for v in kn.split("x")) for kn in os.environ.get("KN", "8192x8192,8192x28672,28672x8192,16384x4096,4096x8192,4096x4096,14336x4096").split(",")]
MS = [int(x) for x in os.environ.get("MS", "1,16,64,128,512,2048").split(",")]
PADS = [0] + [int(x) for x in os.environ.get("PADS", "256").split(",")]
KID = int(os.environ.get("KID", "0"))
for (K, N) in KN:
    for M in MS:
        res = []
        for pad in PADS:
            LD = K + pad
            nb = min(16, max(2, (320 << 20) // (N * LD)))
            Bs = [torch.randint(0, 120, (N, LD), dtype=torch.uint8, device=dev, generator=g) for _ in range(nb)]
            A = torch.randint(0, 120, (M, LD), dtype=torch.uint8, device=dev, generator=g)
            C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            def run(i):
                return lib.fp8mi_scaled_mm_ws(A.data_ptr(), Bs[i % nb].data_ptr(), C.data_ptr(), s1.data_ptr(), s1.data_ptr(), None, None,
                                              M, N, K, LD, LD, N, 0, 0, 2, 0, 0, KID, 0, ws.data_ptr(), ws.numel(), st)
            assert run(0) == 0
            for i in range(nb + 2): run(i)
            torch.cuda.synchronize()
            with L.kernel_timer(24) as kt:
                for i in range(24): run(i)
            torch.cuda.synchronize()
            ms = sorted(kt.ms); res.append(ms[len(ms) // 2] * 1e3)
            kid = lib.fp8mi_choose_kernel(M, N, K, LD, LD, N, 2, 1, 0)
            del Bs, A, C
        print(f"K={K:5d} N={N:5d} M={M:4d} (kernel {kid:2d}): " + "  ".join(f"pad {p}: {t:7.1f}" for p, t in zip(PADS, res)) +
              "   | padded/plain " + " ".join(f"{t / res[0]:.2f}" for t in res[1:]), flush=True)
    torch.cuda.empty_cache()
