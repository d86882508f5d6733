#!/usr/bin/env python3
"""Interleaved A/B of forced kernel ids on one workload in ONE process (medians over rounds).
    python tools/ab_kernels.py <gemm|flux|...> <id> <id> [...]"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import fp8_mi355x_lib as L
name = sys.argv[1]; ids = list(map(int, sys.argv[2:]))
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
ws = {k: bench.Workload(name, dev, kernel=k) for k in ids[:1]}
w = ws[ids[0]]
res = {k: [] for k in ids}
s = torch.cuda.current_stream(dev).cuda_stream
for rnd in range(8):
    for k in ids:
        w.kernel = k
        for i in range(w.inner): w.launch(i, s)          # warm
        torch.cuda.synchronize()
        with L.kernel_timer(2 * w.inner) as kt:
            for i in range(2 * w.inner): w.launch(i, s)
        torch.cuda.synchronize()
        res[k].append(statistics.median(kt.ms) * 1e3)
for k in ids:
    print(f"{name} kernel {k:3d}: median of round-medians {statistics.median(res[k]):8.2f} us   (min {min(res[k]):.2f}, max {max(res[k]):.2f})")
