#!/usr/bin/env python3
"""Interleaved A/B of FP8MI_DEBUG values (diagnostic library: libfp8mi_diag.so reads the variable at every launch) on one
bench workload and kernel id, in ONE process.   FP8MI_LIB_PATH=.../libfp8mi_diag.so python tools/ab_debug.py flux 20 0 2"""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import fp8_mi355x_lib as L
name, kid = sys.argv[1], int(sys.argv[2]); vals = sys.argv[3:]
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
w = bench.Workload(name, dev, kernel=kid)
res = {v: [] for v in vals}
s = torch.cuda.current_stream(dev).cuda_stream
for rnd in range(8):
    for v in vals:
        os.environ["FP8MI_DEBUG"] = v
        for i in range(w.inner): w.launch(i, s)
        torch.cuda.synchronize()
        with L.kernel_timer(2 * w.inner) as kt:
            for i in range(2 * w.inner): w.launch(i, s)
        torch.cuda.synchronize()
        res[v].append(statistics.median(kt.ms) * 1e3)
for v in vals:
    print(f"{name} kernel {kid} FP8MI_DEBUG={v}: median of round-medians {statistics.median(res[v]):8.2f} us   (min {min(res[v]):.2f}, max {max(res[v]):.2f})")
