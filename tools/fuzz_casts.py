#!/usr/bin/env python3
"""Randomised parity stress of the casts (encode / quantize / amax / dequant): sizes, alignments, dtypes.
    python tools/fuzz_casts.py [seed] [cases]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "fp8-mps-metal_amd"), os.path.join(ROOT, "oracle")]
import numpy as np, torch
import fp8_mi355x_native as nat, fp8_oracle as orc
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
dev = torch.device("cuda:0"); rng = np.random.default_rng(seed); bad = 0
for it in range(cases):
    n = int(rng.choice([1, 3, 15, 16, 17, 63, 64, 65, 255, 1000, 1023, 1024, 1025, 4099, 65537, 262147, 1 << 20, (1 << 22) + 5, 3_000_001]))
    off = int(rng.integers(0, 9))
    dt = [torch.float32, torch.float16, torch.bfloat16][int(rng.integers(3))]
    scale = float(rng.choice([1e-4, 0.01, 1.0, 30.0, 500.0]))
    x = (torch.from_numpy(rng.standard_normal(n + off).astype(np.float32)) * scale).to(dt)
    if rng.random() < 0.3: x[rng.integers(0, n + off)] = float(rng.choice([0.0, -0.0, 448.0, -448.0, 1e9, -1e9, 2.0 ** -10, 2.0 ** -9]))
    x = torch.nan_to_num(x, posinf=60000.0, neginf=-60000.0)   # (fp16 overflow of the 1e9 probe: amax = inf is undefined in the reference too)
    xs = x.to(dev)[off:]                      # misaligned view when off % (16 / itemsize) != 0
    ref_in = x[off:].float().numpy()
    # encode
    e = nat.fp8_encode(xs).cpu().numpy(); ee = orc.encode(ref_in)
    # amax / quantize
    am = float(nat.fp8_amax(xs).cpu()); eam = float(np.max(np.abs(ref_in))) if n else 0.0
    q, inv = nat.fp8_quantize(xs); eq, einv = orc.quantize(ref_in)
    # dequant of random bytes
    b = rng.integers(0, 256, size=n + off, dtype=np.uint8)
    d = nat.fp8_dequantize(torch.from_numpy(b).to(dev)[off:]).cpu().view(torch.int16).numpy().view(np.uint16)
    ed = orc.dequantize_f16(b[off:]).view(np.uint16)
    ok = np.array_equal(e, ee) and am == eam and np.array_equal(q.cpu().numpy(), eq) and float(inv.cpu()) == float(einv) and np.array_equal(d, ed)
    if not ok:
        bad += 1
        print(f"FAIL case {it}: n={n} off={off} dt={dt} scale={scale}: encode {np.array_equal(e, ee)} amax {am == eam} ({am} vs {eam}) "
              f"quantize {np.array_equal(q.cpu().numpy(), eq)} inv {float(inv.cpu()) == float(einv)} dequant {np.array_equal(d, ed)}", flush=True)
print(f"seed {seed}: {cases} cases, {bad} failures")
sys.exit(1 if bad else 0)
