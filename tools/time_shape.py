#!/usr/bin/env python3
"""Per-dispatch kernel time of one scaled_mm shape.  python tools/time_shape.py M K N kernel [out: f32|bf16] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fp8-mps-metal_amd")]
import torch, fp8_mi355x_lib as L
M, K, N, kid = map(int, sys.argv[1:5])
out = sys.argv[5] if len(sys.argv) > 5 else "f32"
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 40
nan_mode = int(sys.argv[7]) if len(sys.argv) > 7 else 0
dev = torch.device("cuda:0"); lib = L.load()
g = torch.Generator(device=dev).manual_seed(1)
PAD = int(os.environ.get("PAD", "0"))  # extra bytes per row (row stride K + PAD): power-of-two strides vs cache sets / channels
LD = K + PAD
A = torch.randint(0, 120, (M, LD), dtype=torch.uint8, device=dev, generator=g)
nb = max(2, (320 << 20) // max(N * LD, 1)); nb = min(nb, 24)
nb = int(os.environ.get("NB", nb))  # NB=1: one weight buffer (cache-resident after the warm-up launches)
Bs = [torch.randint(0, 120, (N, LD), dtype=torch.uint8, device=dev, generator=g) for _ in range(nb)]
PADC = int(os.environ.get("PADC", "0"))  # extra ELEMENTS per row of C (ldc = N + PADC)
LDC = N + PADC
C = torch.empty(M, LDC, dtype=torch.float32 if out == "f32" else torch.bfloat16, device=dev)
s1 = torch.full((1,), 0.01, device=dev)
st = torch.cuda.current_stream().cuda_stream
SPLIT = int(os.environ.get("SPLIT", "0"))  # split-K: 0 auto, 1 none, > 1 forced
ws = torch.zeros(int(lib.fp8mi_scaled_mm_workspace_bytes()), dtype=torch.uint8, device=dev)
def run(i):
    L.check(lib.fp8mi_scaled_mm_ws(A.data_ptr(), Bs[i % nb].data_ptr(), C.data_ptr(), s1.data_ptr(), s1.data_ptr(), None, None,
                                   M, N, K, LD, LD, LDC, 0, 0, 0 if out == "f32" else 2, 0, nan_mode, kid, SPLIT,
                                   ws.data_ptr(), ws.numel(), st), "mm")
for i in range(5): run(i)
torch.cuda.synchronize()
with L.kernel_timer(reps) as kt:
    for i in range(reps): run(i)
torch.cuda.synchronize()
ms = sorted(kt.ms)
print(f"M={M} K={K} N={N} pad={PAD} padc={PADC} split={SPLIT} kernel={kid} out={out} nan_mode={nan_mode}: avg {sum(ms)/len(ms)*1e3:.2f} us  min {ms[0]*1e3:.2f}  ({2.0*M*N*K/(sum(ms)/len(ms)*1e-3)/1e12:.1f} TFLOP/s)")
