#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the GEMM K loop (needs libfp8mi_stamp.so).
    FP8MI_LIB_PATH=fp8-mps-metal_amd/libfp8mi_stamp.so python tools/stamp_gemm.py <gemm|flux> <kernel_id>"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
import fp8_mi355x_lib as L
name, kernel = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
w = bench.Workload(name, dev, kernel=kernel)
s = torch.cuda.current_stream(dev).cuda_stream
for i in range(4): w.launch(i, s)
torch.cuda.synchronize()
lib = L.load()
buf = (ctypes.c_ulonglong * (256 * 32))()
lib.fp8mi_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.fp8mi_debug_read_stamps(buf, 256 * 32) == 0
import numpy as np
full = np.array(list(buf), dtype=np.float64).reshape(256, 32)
a = full[:, :8]
nk = a[:, 4].max()
for wname, b in (("wave 0 (loads)", full[:, 0:8]), ("wave kWaves/2", full[:, 8:16])):
    print(f"{name} kernel {kernel}: nk={int(nk)}; mean ticks per K-step over 256 blocks, {wname}:")
    for i, n in enumerate(("wait(vmcnt)", "barrier", "issue loads", "ds_read+MFMA")):
        print(f"  {n:14s} {b[:, i].mean() / nk:8.1f}   (min {b[:, i].min() / nk:7.1f}, max {b[:, i].max() / nk:7.1f})")
    print(f"  total          {b[:, :4].sum(1).mean() / nk:8.1f}")
t0 = full[:, 16:17]
w0, w4 = full[:, 16:21] - t0, full[:, 21:26] - t0
print("step 5, ticks relative to wave 0's loop top (mean over blocks):  top  waited  barrier-passed  issued  MFMAs-done")
print("  wave 0       ", " ".join(f"{x:8.0f}" for x in w0.mean(0)))
print("  wave kWaves/2", " ".join(f"{x:8.0f}" for x in w4.mean(0)))
tot = a[:, 5] + a[:, 6]
print(f"in-kernel shader clock over the tile: {(tot / a[:, 7]).mean() * 0.1:.2f} GHz (min {(tot / a[:, 7]).min() * 0.1:.2f}, max {(tot / a[:, 7]).max() * 0.1:.2f}); tile wall {a[:, 7].mean() / 100:.1f} us")
print(f"per tile (wave 0): entry->end of K loop {a[:, 5].mean():9.0f} ticks (loop body {a[:, :4].sum(1).mean():9.0f}), epilogue+store drain {a[:, 6].mean():9.0f} ticks")
ph = full[:, 26:31]
print("phases of wave 0, ticks (mean / min / max over blocks):")
for name, v in (("entry -> prologue issue starts", ph[:, 0] - ph[:, 3]), ("prologue issue (first stages' DMA)", ph[:, 1] - ph[:, 0]),
                ("K loop incl. final barrier", ph[:, 2] - ph[:, 1]), ("NaN vote / split-K", ph[:, 4] - ph[:, 2])):
    print(f"  {name:36s} {v.mean():9.0f} {v.min():9.0f} {v.max():9.0f}")
