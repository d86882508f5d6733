"""Regret of the shipped dispatch on every committed fixture of measured times - and, optionally, of another build of the library beside it
(e.g. the hand-written rules of round 3:  git archive 5637f2f fp8-mps-metal_amd include | tar -x -C /tmp/oldlib && make -C /tmp/oldlib/fp8-mps-metal_amd).
    python tools/dispatch_fit/compare.py [/tmp/oldlib/fp8-mps-metal_amd/libfp8mi.so]"""
import ctypes, os, statistics, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "fp8-mps-metal_amd"), HERE]
import data
import fp8_mi355x_lib as L
libs = [("cost model (shipped)", L.load())]
if len(sys.argv) > 1:
    old = ctypes.CDLL(sys.argv[1])
    old.fp8mi_choose_kernel.argtypes = [ctypes.c_int64] * 6 + [ctypes.c_int] * 3
    libs.insert(0, ("other build", old))
IDS = {"mx": L.KERNEL_GEMV_MX, "skinny": L.KERNEL_SKINNY, "32x32": L.KERNEL_GEMM_32x32, "32x64": L.KERNEL_GEMM_32x64, "64x64": L.KERNEL_GEMM_64x64,
       "64x128": L.KERNEL_GEMM_64x128, "128x64": L.KERNEL_GEMM_128x64, "128": L.KERNEL_GEMM_128, "128D": L.KERNEL_GEMM_128D,
       "256W": L.KERNEL_GEMM_256W, "256x128W": L.KERNEL_GEMM_256x128W, "gemv": L.KERNEL_GEMV}
NAMES = {v: k for k, v in IDS.items()}
for path, ws in ((data.FIXTURE_FIT, 1), (data.FIXTURE_ANCHORS, 1), (data.FIXTURE_HELDOUT, 1), (data.FIXTURE_NOWS, 0)):
    d = data.load_fixture(path)
    for tag, lib in libs:
        r, miss = [], 0
        for (M, K, N, out), t in d.items():
            pick = NAMES.get(lib.fp8mi_choose_kernel(M, N, K, K, K, N, 0 if out == "f32" else 2, ws, 0))
            if pick not in t:
                miss += 1
                continue
            r.append(t[pick] / min(t.values()))
        print(f"{os.path.basename(path):34s} {tag:22s}: {len(r):4d} shapes ({miss:2d} picks unmeasured)  median {statistics.median(r):.3f}  > 1.05: {sum(x > 1.05 for x in r):3d}  "
              f"> 1.10: {sum(x > 1.10 for x in r):3d}  > 1.20: {sum(x > 1.20 for x in r):3d}")
