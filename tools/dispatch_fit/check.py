"""libfp8mi.so's dispatch (fp8mi_predict_kernel_us / fp8mi_choose_kernel) against the Python twin of the model and against measured times.
    python tools/dispatch_fit/check.py [raw sweep glob] [NOWS]      (default: the committed fixture; NOWS = the sweep ran without a workspace)"""
import json, os, statistics, sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(ROOT, "fp8-mps-metal_amd"), HERE]
from data import load_fixture, load_raw
from model import predict
import fp8_mi355x_lib as L
lib = L.load()
IDS = {"mx": L.KERNEL_GEMV_MX, "skinny": L.KERNEL_SKINNY, "32x32": L.KERNEL_GEMM_32x32, "32x64": L.KERNEL_GEMM_32x64, "64x64": L.KERNEL_GEMM_64x64,
       "64x128": L.KERNEL_GEMM_64x128, "128x64": L.KERNEL_GEMM_128x64, "128": L.KERNEL_GEMM_128, "128D": L.KERNEL_GEMM_128D,
       "256W": L.KERNEL_GEMM_256W, "256x128W": L.KERNEL_GEMM_256x128W, "gemv": L.KERNEL_GEMV}
NAMES = {v: k for k, v in IDS.items()}
args = [a for a in sys.argv[1:] if a != "NOWS"]
ws = 0 if "NOWS" in sys.argv[1:] else 1
data = load_raw(args[0]) if args else load_fixture()
consts = json.load(open(os.path.join(HERE, "constants.json")))
worst, rows, errs = 0.0, [], []
for (M, K, N, out), t in data.items():
    oc, esz = (0, 4) if out == "f32" else (2, 2)
    for k in t:
        if k == "gemv" or M == 1: continue
        cpp = lib.fp8mi_predict_kernel_us(IDS[k], M, N, K, K, K, N, oc, ws, 0, 256)
        py = predict(consts, k, M, N, K, esz, has_ws=bool(ws))
        assert cpp > 0, (k, M, K, N)
        worst = max(worst, abs(cpp - py) / py)
        errs.append(abs(cpp / t[k] - 1))
    pick = NAMES.get(lib.fp8mi_choose_kernel(M, N, K, K, K, N, oc, ws, 0), "?")
    if pick not in t:
        print("pick not measured:", pick, M, K, N, out); continue
    best = min(t, key=t.get)
    rows.append((t[pick] / t[best], M, K, N, out, pick, best, t[pick], t[best]))
print(f"largest relative difference C++ vs Python twin: {worst:.2e};  median |predicted / measured - 1| = {statistics.median(errs):.3f}")
rows.sort(reverse=True)
r = [x[0] for x in rows]
print(f"dispatch on {len(rows)} measured shapes: median regret {statistics.median(r):.3f}  > 1.05: {sum(x > 1.05 for x in r)}  > 1.10: {sum(x > 1.10 for x in r)}  > 1.20: {sum(x > 1.20 for x in r)}")
for x in rows[:12]:
    print(f"  {x[0]:.2f} M={x[1]} K={x[2]} N={x[3]} {x[4]}: pick {x[5]} {x[7]:.1f} us, best {x[6]} {x[8]:.1f} us")
