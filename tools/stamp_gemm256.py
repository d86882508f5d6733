#!/usr/bin/env python3
"""Diagnostic: phase stamps of the hand-scheduled 256x256 kernel (needs libfp8mi_stamp.so).
    FP8MI_LIB_PATH=fp8-mps-metal_amd/libfp8mi_stamp.so python tools/stamp_gemm256.py <flux|...> [kernel_id=20]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, bench
import fp8_mi355x_lib as L
name = sys.argv[1]; kernel = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
w = bench.Workload(name, dev, kernel=kernel)
s = torch.cuda.current_stream(dev).cuda_stream
for i in range(4): w.launch(i, s)
torch.cuda.synchronize()
lib = L.load()
buf = (ctypes.c_ulonglong * (1024 * 8))()
lib.fp8mi_debug_read_stamps256.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.fp8mi_debug_read_stamps256(buf, 1024 * 8) == 0
a = np.array(list(buf), dtype=np.float64).reshape(1024, 8)
a = a[a[:, 7] > 0]
print(f"{name}: {len(a)} workgroups stamped")
t_first = a[:, 6].min()
names = ("entry -> K loop (setup)", "K loop (asm: prologue DMA .. dump of rows 0-3)", "epilogue, fragment rows 0-3 (from the dump)", "dump + epilogue, fragment rows 4-7", "store drain (vmcnt 0)")
wall = (a[:, 7] - a[:, 6]) / 100.0   # us (100 MHz)
clk = (a[:, 5] - a[:, 0]) / np.maximum(a[:, 7] - a[:, 6], 1) * 0.1
print(f"shader clock {clk.mean():.2f} GHz; workgroup wall {wall.mean():.1f} us (min {wall.min():.1f}, max {wall.max():.1f})")
for i, n in enumerate(names):
    d = (a[:, i + 1] - a[:, i])
    print(f"  {n:42s} {d.mean():9.0f} cycles  = {d.mean() / clk.mean() / 1e3:6.2f} us   (min {d.min():.0f}, max {d.max():.0f})")
start = (a[:, 6] - t_first) / 100.0; end = (a[:, 7] - t_first) / 100.0
order = np.argsort(start)
q = len(a) // 3 if len(a) >= 768 else len(a)
for r in range(0, len(a), max(q, 1)):
    sel = order[r:r + q]
    print(f"  workgroups {r:4d}..{r + len(sel) - 1:4d} (by start): start {start[sel].mean():7.1f} us (min {start[sel].min():.1f} max {start[sel].max():.1f})  end {end[sel].mean():7.1f} (min {end[sel].min():.1f} max {end[sel].max():.1f})")
