#!/usr/bin/env python3
"""Launch one bench workload eagerly N times (for rocprofv3 --pmc passes, where a
plain launch stream is easier to read than a graph replay).
    python tools/run_workload.py <gemm|gemv|flux|quantize|dequant> [launches] [kernel_id]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

name = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
kernel = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
w = bench.Workload(name, dev, kernel=kernel)
s = torch.cuda.current_stream(dev).cuda_stream
for i in range(n):
    w.launch(i, s)
torch.cuda.synchronize()
print(f"{name}: {n} launches done")
