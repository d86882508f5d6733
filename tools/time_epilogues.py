#!/usr/bin/env python3
"""Per-dispatch time of one shape through AUTO with each epilogue form (per-tensor / per-row scales, bias, scale_result).
    python tools/time_epilogues.py M K N [f32|bf16]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fp8-mps-metal_amd")]
import torch, fp8_mi355x_lib as L, fp8_mi355x_native as native
M, K, N = map(int, sys.argv[1:4]); od = torch.float32 if (len(sys.argv) > 4 and sys.argv[4] == "f32") else torch.bfloat16
dev = torch.device("cuda:0"); g = torch.Generator(device=dev).manual_seed(1)
A = torch.randint(0, 120, (M, K), dtype=torch.uint8, device=dev, generator=g)
nb = max(2, min(12, (320 << 20) // (N * K)))
Bs = [torch.randint(0, 120, (N, K), dtype=torch.uint8, device=dev, generator=g) for _ in range(nb)]
s1 = torch.full((1,), 0.01, device=dev); sM = torch.full((M,), 0.01, device=dev); sN = torch.full((N,), 0.01, device=dev)
bias = torch.randn(N, device=dev).to(od); sr = torch.full((1,), 0.5, device=dev); out = torch.empty(M, N, dtype=od, device=dev)
forms = {"per-tensor": dict(scale_a=s1, scale_b=s1), "per-tensor + bias": dict(scale_a=s1, scale_b=s1, bias=bias),
         "per-row scales": dict(scale_a=sM, scale_b=sN), "per-row + bias": dict(scale_a=sM, scale_b=sN, bias=bias),
         "per-tensor + bias + scale_result": dict(scale_a=s1, scale_b=s1, bias=bias, scale_result=sr)}
for name, kw in forms.items():
    run = lambda i: native.fp8_scaled_mm(A, Bs[i % nb], out_dtype=od, out=out, **kw)
    for i in range(nb + 2): run(i)
    torch.cuda.synchronize()
    with L.kernel_timer(40) as kt:
        for i in range(40): run(i)
    torch.cuda.synchronize()
    ms = sorted(kt.ms)
    print(f"M={M} K={K} N={N} {od}: {name:34s} median {ms[len(ms) // 2] * 1e3:8.2f} us  min {ms[0] * 1e3:8.2f}")
