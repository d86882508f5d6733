#!/usr/bin/env python3
"""Parity of the few-rows kernel (FP8MI_KERNEL_GEMV_MX, 2 <= M <= 8) against the oracle, then timing against the other
small-M kernels.   python tools/check_gemv_mx.py [time]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fp8-mps-metal_amd"), os.path.join(ROOT, "oracle")]
import fp8_mi355x_native as n, fp8_oracle as o, fp8_mi355x_lib as L
dev = torch.device("cuda:0")
worst = 0.0
for M in (2, 3, 4, 5, 7, 8):
    for (K, N) in ((16, 1), (272, 70), (1040, 33), (4096, 513), (8192, 64), (14336, 257), (16384, 40)):
        rng = np.random.default_rng(M * 1000 + K + N)
        for nan in (False, True):
            X = rng.integers(0, 256, size=(M, K), dtype=np.uint8); W = rng.integers(0, 256, size=(N, K), dtype=np.uint8)
            if not nan:
                X[(X & 0x7F) == 0x7F] ^= 1; W[(W & 0x7F) == 0x7F] ^= 1
            else:
                X[M - 1, K // 2] = 0x7F; W[-1, -1] = 0xFF; W[0, 0] = 0x7F
            sa = rng.uniform(0.005, 0.02, size=M).astype(np.float32); sb = rng.uniform(0.005, 0.02, size=N).astype(np.float32)
            bias = rng.standard_normal(N).astype(np.float32)
            got = n.fp8_scaled_mm(torch.from_numpy(X).to(dev), torch.from_numpy(W).to(dev), torch.from_numpy(sa), torch.from_numpy(sb),
                                  bias=torch.from_numpy(bias).to(dev), kernel=L.KERNEL_GEMV_MX)
            torch.cuda.synchronize()
            ex = o.scaled_mm(X, W, sa, sb, accumulate="f64") + bias[None, :]
            bd = o.abs_dot_bound(X, W, sa, sb) + np.abs(bias)[None, :]
            r = float(np.max(np.abs(got.cpu().numpy() - ex) / (bd + 1e-300)))
            worst = max(worst, r)
            if not r <= 1e-3:
                print(f"M={M} K={K} N={N} nan={nan}: max err/bound {r:.3e}  FAIL"); sys.exit(1)
rng = np.random.default_rng(77)
X = (0x28 + rng.integers(0, 0x20, size=(5, 6144))).astype(np.uint8); W = (0x28 + rng.integers(0, 0x20, size=(300, 6144))).astype(np.uint8)
got = n.fp8_scaled_mm(torch.from_numpy(X).to(dev), torch.from_numpy(W).to(dev), torch.ones(1), torch.ones(1), kernel=L.KERNEL_GEMV_MX,
                      out_dtype=torch.float32).cpu().numpy()
ex = o.scaled_mm(X, W, [1.0], [1.0], accumulate="f64")
r = float(np.max(np.abs(got - ex) / o.abs_dot_bound(X, W, [1.0], [1.0])))
print(f"few-rows kernel: all shapes ok (worst err/bound {worst:.2e}; narrow-range {r:.2e})")
assert r <= 4e-6
if len(sys.argv) > 1:
    lib = L.load(); st = torch.cuda.current_stream().cuda_stream
    ws = torch.zeros(int(lib.fp8mi_scaled_mm_workspace_bytes()), dtype=torch.uint8, device=dev)
    s1 = torch.full((1,), 0.01, device=dev)
    g = torch.Generator(device=dev).manual_seed(1)
    for (K, N) in ((4096, 4096), (14336, 4096), (4096, 14336), (8192, 8192), (3072, 12288), (14336, 14336), (8192, 14336), (12288, 3072), (2048, 2048), (4096, 1024)):
        nb = max(2, min(24, (320 << 20) // (N * K)))
        Bs = [torch.randint(0, 120, (N, K), dtype=torch.uint8, device=dev, generator=g) for _ in range(nb)]
        for M in (2, 3, 4, 6, 8):
            A = torch.randint(0, 120, (M, K), dtype=torch.uint8, device=dev, generator=g)
            C = torch.empty(M, N, dtype=torch.float32, device=dev)
            line = f"M={M} K={K} N={N}:"
            cands = [("auto", 0, 0), ("mx", L.KERNEL_GEMV_MX, 1), ("skinny", L.KERNEL_SKINNY, 1), ("64x128+sk", L.KERNEL_GEMM_64x128, 0), ("128x64", L.KERNEL_GEMM_128x64, 1)]
            if "diag" in os.environ.get("FP8MI_LIB_PATH", ""):
                cands = [("g1", 70, 1), ("g2", 71, 1), ("g4", 72, 1), ("g8", 73, 1)] + cands
            for name, kid, split in cands:
                def run(i):
                    L.check(lib.fp8mi_scaled_mm_ws(A.data_ptr(), Bs[i % nb].data_ptr(), C.data_ptr(), s1.data_ptr(), s1.data_ptr(), None, None,
                                                   M, N, K, K, K, N, 0, 0, 0, 0, 0, kid, split, ws.data_ptr(), ws.numel(), st), "mm")
                for i in range(nb): run(i)
                torch.cuda.synchronize()
                with L.kernel_timer(40) as kt:
                    for i in range(40): run(i)
                torch.cuda.synchronize()
                ms = sorted(kt.ms); line += f"  {name} {ms[len(ms)//2]*1e3:6.2f}"
            print(line + "  us (median)")
