import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fp8-mps-metal_amd"), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")]
import numpy as np, torch
import fp8_mi355x_native as nat, fp8_mi355x_lib as L, fp8_oracle as orc
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
def clean(shape):
    b = rng.integers(0, 256, size=shape, dtype=np.uint8); b[(b & 0x7F) == 0x7F] = 0x3C; return b
ok = True
for (M, K, N, kern, split) in [(128, 4096, 512, 5, 4), (64, 2048, 256, 14, 0), (100, 3000 // 16 * 16, 200, 5, 3), (33, 14336, 4096, 0, 0),
                               (128, 14336, 4096, 0, 0), (256, 4096, 1024, 0, 0), (40, 1040, 130, 14, 5), (512, 4096, 4096, 2, 2),
                               (130, 4096, 70, 5, 16), (64, 8192, 128, 14, 16)]:
    A, B = clean((M, K)), clean((N, K))
    sa = rng.uniform(0.005, 0.02, size=M).astype(np.float32); sb = rng.uniform(0.005, 0.02, size=N).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    At, Bt = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    for od in (torch.float32, torch.bfloat16):
        got = nat.fp8_scaled_mm(At, Bt, torch.from_numpy(sa), torch.from_numpy(sb), bias=torch.from_numpy(bias), out_dtype=od, kernel=kern, split_k=split)
        got2 = nat.fp8_scaled_mm(At, Bt, torch.from_numpy(sa), torch.from_numpy(sb), bias=torch.from_numpy(bias), out_dtype=od, kernel=kern, split_k=split)
        same = torch.equal(got, got2)
        ref = nat.fp8_scaled_mm(At, Bt, torch.from_numpy(sa), torch.from_numpy(sb), bias=torch.from_numpy(bias), out_dtype=od, kernel=kern, split_k=1)
        exact = orc.scaled_mm(A, B, sa, sb, accumulate="f64") + bias[None, :]
        bound = orc.abs_dot_bound(A, B, sa, sb)
        err = np.abs(got.float().cpu().numpy().astype(np.float64) - exact)
        tol = 1e-3 * bound + (np.abs(exact) * 2.0 ** -8 if od == torch.bfloat16 else 0) + 1e-30
        good = bool(np.all(err <= tol))
        dref = (got.float() - ref.float()).abs().max().item()
        print(f"M={M} K={K} N={N} kern={kern} split={split} {str(od)[6:]:9s}: parity {'ok' if good else 'FAIL'} worst err/bound {np.max(err / (bound + 1e-300)):.2e}  reproducible {same}  max|split - nosplit| {dref:.3e}")
        ok &= good and same
ws = nat._workspace(dev)
print("counters zero after use:", int(ws[:4096].view(torch.int32).abs().sum().item()) == 0)
print("ALL OK" if ok else "FAILURES")
