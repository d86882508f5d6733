#!/usr/bin/env python3
"""Where the host time of one eager torch._scaled_mm call goes (patched surface, float8 tensors, device scales).
    python tools/time_callsite.py [M K N]      -> per-call issue cost of the layers, then a cProfile of 3000 calls"""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fp8-mps-metal_amd")]
import torch
import fp8_mi355x_lib as L, fp8_mi355x_native as native, fp8_mps_patch

M, K, N = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (1, 4096, 4096)
dev = torch.device("cuda:0")
a8 = torch.randint(0, 120, (M, K), dtype=torch.uint8, device=dev)
b8 = torch.randint(0, 120, (N, K), dtype=torch.uint8, device=dev)
A, Bt = a8.view(torch.float8_e4m3fn), b8.view(torch.float8_e4m3fn).t()
sa = torch.full((1,), 0.01, device=dev); sb = sa.clone()
C = torch.empty(M, N, device=dev)
lib = L.load()
st = torch.cuda.current_stream().cuda_stream


def per_call(fn, n=3000):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6, (time.perf_counter() - t0) / n * 1e6


def raw():
    lib.fp8mi_scaled_mm_ws(a8.data_ptr(), b8.data_ptr(), C.data_ptr(), sa.data_ptr(), sb.data_ptr(), None, None, M, N, K, K, K, N,
                           0, 0, 0, 0, 0, 0, 1, None, 0, st)


fp8_mps_patch.install()
rows = [("ctypes call only (preallocated output)", raw),
        ("torch.empty((M,N))", lambda: torch.empty((M, N), dtype=torch.float32, device=dev)),
        ("native.fp8_scaled_mm (uint8, (N,K))", lambda: native.fp8_scaled_mm(a8, b8, sa, sb)),
        ("native.scaled_mm_colmajor (fp8, (K,N))", lambda: native.scaled_mm_colmajor(A, Bt, sa, sb)),
        ("patched torch._scaled_mm", lambda: torch._scaled_mm(A, Bt, scale_a=sa, scale_b=sb, out_dtype=torch.float32))]
for name, fn in rows:
    issue, total = per_call(fn)
    print(f"{name:44s} issue {issue:6.2f} us/call   issue+drain {total:6.2f} us/call")
pr = cProfile.Profile()
pr.enable()
for _ in range(3000):
    torch._scaled_mm(A, Bt, scale_a=sa, scale_b=sb, out_dtype=torch.float32)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
fp8_mps_patch.uninstall()
