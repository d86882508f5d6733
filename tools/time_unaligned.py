#!/usr/bin/env python3
"""Op-level fp8_scaled_mm on operands the MFMA kernels cannot read in place (K % 16 != 0; a sliced, misaligned weight view): device time per call
of the whole op (pad copies + kernel) against the forced generic kernel and against the aligned neighbour shape.  -> profiles/r04_unaligned.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fp8-mps-metal_amd")]
import torch, fp8_mi355x_lib as L, fp8_mi355x_native as native
dev = torch.device("cuda:0"); L.load()
g = torch.Generator(device=dev).manual_seed(1)
one = torch.full((1,), 0.01, device=dev)


def timed(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for (M, K, N) in ((4096, 4100, 4096), (512, 4100, 4096), (64, 4100, 4096), (1, 4100, 4096)):
    A = torch.randint(0, 120, (M, K), dtype=torch.uint8, device=dev, generator=g)
    B = torch.randint(0, 120, (N, K), dtype=torch.uint8, device=dev, generator=g)
    Ka = K // 16 * 16
    Aa, Ba = A[:, :Ka].contiguous(), B[:, :Ka].contiguous()
    t_pad = timed(lambda: native.fp8_scaled_mm(A, B, one, one, out_dtype=torch.bfloat16), 20)
    t_gen = timed(lambda: native.fp8_scaled_mm(A, B, one, one, out_dtype=torch.bfloat16, kernel=L.KERNEL_GENERIC), 2 if M >= 512 else 10)
    t_al = timed(lambda: native.fp8_scaled_mm(Aa, Ba, one, one, out_dtype=torch.bfloat16), 20)
    print(f"M={M} K={K} N={N}: op with padded copies {t_pad:9.1f} us | generic kernel (round 3's path) {t_gen:10.1f} us | aligned K={Ka} {t_al:8.1f} us | speed-up {t_gen / t_pad:7.1f}x")
big = torch.randint(0, 120, (4104, 4120), dtype=torch.uint8, device=dev, generator=g)
W = big[8:, 8:8 + 4096]     # (4096, 4096) view: base + 8 bytes, row stride 4120
X = torch.randint(0, 120, (512, 4096), dtype=torch.uint8, device=dev, generator=g)
t_pad = timed(lambda: native.fp8_scaled_mm(X, W, one, one, out_dtype=torch.bfloat16), 20)
t_gen = timed(lambda: native.fp8_scaled_mm(X, W, one, one, out_dtype=torch.bfloat16, kernel=L.KERNEL_GENERIC), 3)
print(f"sliced weight view (4096, 4096), base % 16 = {W.data_ptr() % 16}, row stride {W.stride(0)}; M=512: op with aligned copy {t_pad:9.1f} us | generic kernel {t_gen:10.1f} us | speed-up {t_gen / t_pad:7.1f}x")
