#!/usr/bin/env python3
"""Parity of forced M == 1 kernel ids against the oracle (ragged K, NaN bytes, narrow-range exactness).
    python tools/check_gemv.py <kernel_id> [...]      (diagnostic ids need FP8MI_LIB_PATH=.../libfp8mi_diag.so)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fp8-mps-metal_amd"), os.path.join(ROOT, "oracle")]
import fp8_mi355x_native as n, fp8_oracle as o
dev = torch.device("cuda:0")
shapes = [(16, 1), (128, 5), (272, 70), (1040, 33), (4096, 512), (4112, 100), (8192, 64), (14336, 257), (20480, 40), (32768, 9)]
for kid in map(int, sys.argv[1:]):
    worst = 0.0
    for (K, N) in shapes:
        rng = np.random.default_rng(K + N)
        for nan in (False, True):
            x = rng.integers(0, 256, size=(1, K), dtype=np.uint8); W = rng.integers(0, 256, size=(N, K), dtype=np.uint8)
            if not nan:
                x[(x & 0x7F) == 0x7F] ^= 1; W[(W & 0x7F) == 0x7F] ^= 1
            else:
                x[0, K // 2] = 0x7F; W[-1, -1] = 0xFF; W[0, 0] = 0x7F
            sb = rng.uniform(0.005, 0.02, size=N).astype(np.float32); bias = rng.standard_normal(N).astype(np.float32)
            got = n.fp8_scaled_mm(torch.from_numpy(x).to(dev), torch.from_numpy(W).to(dev), torch.tensor([0.013]), torch.from_numpy(sb),
                                  bias=torch.from_numpy(bias).to(dev), kernel=kid)
            torch.cuda.synchronize()
            ex = o.scaled_mm(x, W, [0.013], sb, accumulate="f64") + bias[None, :]
            bd = o.abs_dot_bound(x, W, [0.013], sb) + np.abs(bias)[None, :]
            r = float(np.max(np.abs(got.cpu().numpy() - ex) / (bd + 1e-300)))
            worst = max(worst, r)
            if not r <= 1e-3:
                print(f"kernel {kid} K={K} N={N} nan={nan}: max err/bound {r:.3e}  FAIL"); sys.exit(1)
    rng = np.random.default_rng(77)
    x = (0x28 + rng.integers(0, 0x20, size=(1, 6144))).astype(np.uint8); W = (0x28 + rng.integers(0, 0x20, size=(300, 6144))).astype(np.uint8)
    got = n.fp8_scaled_mm(torch.from_numpy(x).to(dev), torch.from_numpy(W).to(dev), torch.ones(1), torch.ones(1), kernel=kid).cpu().numpy()
    ex = o.scaled_mm(x, W, [1.0], [1.0], accumulate="f64")
    r = float(np.max(np.abs(got - ex) / o.abs_dot_bound(x, W, [1.0], [1.0])))
    print(f"kernel {kid}: all shapes ok (worst err/bound {worst:.2e}; narrow-range {r:.2e})")
    assert r <= 4e-6
