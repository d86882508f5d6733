#!/usr/bin/env python3
"""Parity of one forced GEMM kernel id against the oracle on a set of shapes.
    python tools/check_kernel.py <kernel_id> [...]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fp8-mps-metal_amd"), os.path.join(ROOT, "oracle")]
import fp8_mi355x_native as n, fp8_oracle as o
dev = torch.device("cuda:0")
shapes = [(1, 128, 1), (5, 16, 3), (128, 128, 128), (130, 272, 70), (300, 1040, 200), (256, 512, 256), (257, 384, 513),
          (64, 4096, 96), (512, 4096, 512), (1024, 3072, 768)]
for kid in map(int, sys.argv[1:]):
    worst = 0.0
    for (M, K, N) in shapes:
        rng = np.random.default_rng(M + K + N)
        A = rng.integers(0, 256, size=(M, K), dtype=np.uint8); B = rng.integers(0, 256, size=(N, K), dtype=np.uint8)
        for nan in (False, True):
            if not nan:
                A[(A & 0x7F) == 0x7F] ^= 1; B[(B & 0x7F) == 0x7F] ^= 1
            else:
                A[0, 0] = 0x7F; B[-1, -1] = 0xFF
            sa = rng.uniform(0.005, 0.02, size=M).astype(np.float32); sb = rng.uniform(0.005, 0.02, size=N).astype(np.float32)
            got = n.fp8_scaled_mm(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), torch.from_numpy(sa), torch.from_numpy(sb), kernel=kid)
            torch.cuda.synchronize()
            ex = o.scaled_mm(A, B, sa, sb, accumulate="f64"); bd = o.abs_dot_bound(A, B, sa, sb)
            r = float(np.max(np.abs(got.cpu().numpy() - ex) / (bd + 1e-300)))
            worst = max(worst, r)
            if not r <= 1e-3:
                print(f"kernel {kid} shape {(M, K, N)} nan={nan}: max err/bound {r:.3e}  FAIL"); sys.exit(1)
    # narrow-range exactness
    rng = np.random.default_rng(77)
    A = (0x28 + rng.integers(0, 0x20, size=(200, 1024))).astype(np.uint8); B = (0x28 + rng.integers(0, 0x20, size=(328, 1024))).astype(np.uint8)
    got = n.fp8_scaled_mm(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), torch.ones(1), torch.ones(1), kernel=kid).cpu().numpy()
    ex = o.scaled_mm(A, B, [1.0], [1.0], accumulate="f64")
    r = float(np.max(np.abs(got - ex) / o.abs_dot_bound(A, B, [1.0], [1.0])))
    print(f"kernel {kid}: all shapes ok (worst err/bound {worst:.2e}; narrow-range {r:.2e})")
    assert r <= 4e-6
