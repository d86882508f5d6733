#!/usr/bin/env python3
"""Randomised parity stress of fp8_scaled_mm through the automatic dispatch (all kernels, split-K on / off / forced).
    python tools/fuzz_mm.py [seed] [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fp8-mps-metal_amd"), os.path.join(ROOT, "oracle")]
import numpy as np, torch
import fp8_mi355x_native as nat, fp8_mi355x_lib as L, fp8_oracle as orc
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0"); rng = np.random.default_rng(seed)
pick = lambda xs: xs[int(rng.integers(len(xs)))]
bad = 0
for it in range(cases):
    M = int(pick([1, 1, 2, 3, 4, 7, 8, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 96, 127, 128, 129, 200, 255, 256, 257, 300, 384, 511, 512, 700, 1024]))
    N = int(pick([1, 3, 15, 16, 17, 63, 64, 65, 100, 127, 128, 129, 255, 256, 257, 384, 500, 512, 777, 1024, 2048, 3000, 4096, 8192, 12288]))
    K = int(pick([16, 32, 48, 112, 128, 144, 240, 256, 272, 512, 528, 1024, 1040, 2048, 3072, 4096, 4112, 6144, 8192, 14336]))
    if rng.random() < 0.08: K = int(rng.integers(1, 400))
    while M * N * K > 1.5e9: K = max(16, K // 2)
    A = rng.integers(0, 256, size=(M, K), dtype=np.uint8); B = rng.integers(0, 256, size=(N, K), dtype=np.uint8)
    if rng.random() < 0.8:  # mostly clean bytes; sometimes NaN patterns stay in (reference mode: decode to 0)
        A[(A & 0x7F) == 0x7F] ^= 1; B[(B & 0x7F) == 0x7F] ^= 1
    sa = rng.uniform(0.005, 0.02, size=M if rng.random() < 0.5 else 1).astype(np.float32)
    sb = rng.uniform(0.005, 0.02, size=N if rng.random() < 0.5 else 1).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32) if rng.random() < 0.4 else None
    od = pick([torch.float32, torch.bfloat16, torch.float16]); split = int(pick([0, 0, 0, 1, 2, 3, 4, 7, 16]))
    got = nat.fp8_scaled_mm(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev), torch.from_numpy(sa), torch.from_numpy(sb),
                            bias=None if bias is None else torch.from_numpy(bias), out_dtype=od, split_k=split)
    exact = orc.scaled_mm(A, B, sa, sb, accumulate="f64"); bound = orc.abs_dot_bound(A, B, sa, sb)
    if bias is not None: exact = exact + bias[None, :].astype(np.float64); bound = bound + np.abs(bias)[None, :]
    eps = {torch.float32: 0.0, torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11}[od]
    g = got.float().cpu().numpy().astype(np.float64)
    tiny = 2.0 ** -24 if od == torch.float16 else 0.0   # fp16 results below 6e-5 are subnormal: absolute quantum 2^-24
    ok = np.all(np.abs(g - exact) <= 1e-3 * bound + eps * np.abs(exact) + tiny + 1e-30)
    if not ok:
        bad += 1
        print(f"FAIL case {it}: M={M} K={K} N={N} sa={sa.size} sb={sb.size} bias={bias is not None} out={od} split={split} "
              f"worst {np.max(np.abs(g - exact) / (bound + 1e-300)):.3e}", flush=True)
ws = nat._workspace(dev); torch.cuda.synchronize()
print(f"seed {seed}: {cases} cases, {bad} failures; counters zero: {int(ws[:4096].view(torch.int32).abs().sum().item()) == 0}")
sys.exit(1 if bad else 0)
