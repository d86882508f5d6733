#!/usr/bin/env python3
"""Randomised check of the automatic dispatch on LARGE shapes (where it picks the one-wave-per-SIMD kernels): the result must
equal the unsplit ring kernel's bit for bit, and a sampled block must meet the oracle's bound.
    python tools/fuzz_large.py [seed] [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fp8-mps-metal_amd"), os.path.join(ROOT, "oracle")]
import numpy as np, torch
import fp8_mi355x_native as nat, fp8_mi355x_lib as L, fp8_oracle as orc
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = torch.device("cuda:0"); rng = np.random.default_rng(seed); g = torch.Generator(device=dev).manual_seed(seed)
bad = 0
for it in range(cases):
    M = int(rng.integers(700, 6000)); N = 8 * int(rng.integers(100, 1200)); K = 128 * int(rng.integers(2, 40)) + (16 * int(rng.integers(1, 8)) if rng.random() < 0.3 else 0)   # (round 3: K tails too)
    A = torch.randint(0, 0x7F, (M, K), dtype=torch.uint8, device=dev, generator=g)
    B = torch.randint(0, 0x7F, (N, K), dtype=torch.uint8, device=dev, generator=g)
    if rng.random() < 0.3: A[int(rng.integers(M)), int(rng.integers(K))] = 0x7F          # NaN byte: scrubbing redo of one tile
    sa = torch.from_numpy(rng.uniform(0.005, 0.02, size=M if rng.random() < 0.5 else 1).astype(np.float32)).to(dev)
    sb = torch.from_numpy(rng.uniform(0.005, 0.02, size=N if rng.random() < 0.5 else 1).astype(np.float32)).to(dev)
    bias = torch.from_numpy(rng.normal(size=N).astype(np.float32)).to(dev) if rng.random() < 0.5 else None
    od = [torch.float32, torch.bfloat16, torch.float16][int(rng.integers(3))]
    got = nat.fp8_scaled_mm(A, B, sa, sb, bias=bias, out_dtype=od, split_k=1)   # (unsplit: an automatic K split reorders the sum)
    ref = nat.fp8_scaled_mm(A, B, sa, sb, bias=bias, out_dtype=od, kernel=L.KERNEL_GEMM_256, split_k=1)
    same = torch.equal(got, ref)
    r0, c0 = int(rng.integers(0, M - 64)), int(rng.integers(0, N - 64))
    a, b = A[r0:r0 + 64].cpu().numpy(), B[c0:c0 + 64].cpu().numpy()
    san = sa.cpu().numpy(); sbn = sb.cpu().numpy()
    san = san[r0:r0 + 64] if san.size > 1 else san; sbn = sbn[c0:c0 + 64] if sbn.size > 1 else sbn
    exact = orc.scaled_mm(a, b, san, sbn, accumulate="f64"); bound = orc.abs_dot_bound(a, b, san, sbn)
    if bias is not None:
        bb = bias[c0:c0 + 64].cpu().numpy(); exact = exact + bb[None, :]; bound = bound + np.abs(bb)[None, :]
    gg = got[r0:r0 + 64, c0:c0 + 64].float().cpu().numpy()
    eps = {torch.float32: 0.0, torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[od]
    ok = bool(np.all(np.abs(gg - exact) <= 1e-3 * bound + eps * np.abs(exact) + 1e-6))
    if not (same and ok):
        bad += 1
        print(f"FAIL case {it}: M={M} K={K} N={N} out={od} bit-equal {same} oracle {ok}", flush=True)
print(f"seed {seed}: {cases} large cases, {bad} failures")
