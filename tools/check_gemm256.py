#!/usr/bin/env python3
"""Parity of the hand-scheduled 256x256 kernel (FP8MI_KERNEL_GEMM_256W) against the oracle and - bit for bit - against
the ring kernel on the same inputs.   python tools/check_gemm256.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fp8-mps-metal_amd"), os.path.join(ROOT, "oracle")]
import fp8_mi355x_native as n, fp8_oracle as o, fp8_mi355x_lib as L
dev = torch.device("cuda:0")
KID = int(sys.argv[1]) if len(sys.argv) > 1 else L.KERNEL_GEMM_256W   # a schedule variant of the diagnostic library: 80 + variant
worst = 0.0
for (M, K, N) in ((256, 256, 256), (256, 384, 512), (512, 1024, 256), (768, 640, 1024), (256, 4096, 256), (1024, 3072, 2048),
                  (300, 512, 256), (257, 384, 512), (1, 256, 256), (1000, 640, 768), (129, 1024, 256), (2049, 256, 512),
                  (256, 384, 264), (512, 512, 1000), (300, 640, 520), (64, 1024, 8), (515, 256, 776)):   # ragged N too (a multiple of 8)   # ragged M: last m-tile partly / one wave tile empty
    rng = np.random.default_rng(M + K + N)
    for nan in (False, True):
        A = rng.integers(0, 256, size=(M, K), dtype=np.uint8); B = rng.integers(0, 256, size=(N, K), dtype=np.uint8)
        if not nan:
            A[(A & 0x7F) == 0x7F] ^= 1; B[(B & 0x7F) == 0x7F] ^= 1
        sa = rng.uniform(0.005, 0.02, size=M).astype(np.float32); sb = rng.uniform(0.005, 0.02, size=N).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        tA, tB = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
        for od in (torch.float32, torch.bfloat16):
            got = n.fp8_scaled_mm(tA, tB, torch.from_numpy(sa), torch.from_numpy(sb), bias=torch.from_numpy(bias).to(dev), out_dtype=od, kernel=KID)
            ref = n.fp8_scaled_mm(tA, tB, torch.from_numpy(sa), torch.from_numpy(sb), bias=torch.from_numpy(bias).to(dev), out_dtype=od, kernel=L.KERNEL_GEMM_256, split_k=1)
            torch.cuda.synchronize()
            same = torch.equal(got, ref)
            ex = o.scaled_mm(A, B, sa, sb, accumulate="f64") + bias[None, :]; bd = o.abs_dot_bound(A, B, sa, sb) + np.abs(bias)[None, :]
            eps = 2.0 ** -8 if od == torch.bfloat16 else 0.0
            err = np.abs(got.float().cpu().numpy() - ex)
            r = float(np.max((err - eps * np.abs(ex)) / (bd + 1e-300)))
            worst = max(worst, r)
            print(f"M={M} K={K} N={N} nan={nan} {od}: err/bound {r:.2e}  bit-equal to the ring kernel: {same}")
            if not (r <= 1e-3 and same):
                d = (got.float() - ref.float()).abs().cpu().numpy()
                bad = np.argwhere(d > 0)
                print("  first mismatches (m, n):", bad[:8].tolist(), " count", len(bad))
                sys.exit(1)
print(f"256W (kernel id {KID}): all shapes ok (worst err/bound {worst:.2e})")
