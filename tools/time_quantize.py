#!/usr/bin/env python3
"""Per-dispatch kernel times of fp8mi_quantize (amax + encode) / fp8mi_encode / fp8mi_dequant at one size.
    python tools/time_quantize.py <elements> [f32|f16|bf16]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fp8-mps-metal_amd")]
import torch, fp8_mi355x_lib as L
n = int(sys.argv[1]); dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dt]; code = {"f32": L.F32, "f16": L.F16, "bf16": L.BF16}[dt]
dev = torch.device("cuda:0"); lib = L.load(); st = torch.cuda.current_stream().cuda_stream
nb = max(2, min(16, (512 << 20) // (n * tdt.itemsize)))
xs = [(torch.randn(n, device=dev) * 3).to(tdt) for _ in range(nb)]
q = torch.empty(n, dtype=torch.uint8, device=dev); sc = torch.empty(2, device=dev); h = torch.empty(n, dtype=torch.float16, device=dev)
def stats(fn, per):
    for i in range(nb + 2): fn(i)
    torch.cuda.synchronize()
    with L.kernel_timer(per * 20) as kt:
        for i in range(20): fn(i)
    torch.cuda.synchronize()
    ms = kt.ms
    return [sorted(ms[j::per])[len(ms[j::per]) // 2] * 1e3 for j in range(per)]
t = stats(lambda i: L.check(lib.fp8mi_quantize(xs[i % nb].data_ptr(), code, q.data_ptr(), sc.data_ptr(), n, 0, st), "q"), 2)
esz = tdt.itemsize
print(f"quantize n={n} {dt}: amax {t[0]:.2f} us ({n * esz / t[0] / 1e3:.0f} GB/s)  encode {t[1]:.2f} us ({n * (esz + 1) / t[1] / 1e3:.0f} GB/s)")
t = stats(lambda i: L.check(lib.fp8mi_encode(xs[i % nb].data_ptr(), code, q.data_ptr(), None, n, 0, st), "e"), 1)
print(f"encode   n={n} {dt}: {t[0]:.2f} us ({n * (esz + 1) / t[0] / 1e3:.0f} GB/s)")
t = stats(lambda i: L.check(lib.fp8mi_dequant(q.data_ptr(), h.data_ptr(), None, n, L.F16, st), "d"), 1)
print(f"dequant  n={n} -> f16: {t[0]:.2f} us ({n * 3 / t[0] / 1e3:.0f} GB/s)")
