"""Fits the constants of the dispatch's cost model (model.py = the Python twin of csrc/fp8mi_dispatch.h) to measured per-kernel times and writes
constants.json; emit.py turns that into csrc/fp8mi_dispatch_constants.inc.

    python tools/dispatch_fit/fit.py [raw sweep glob]        (default: the committed fixture tests/golden/dispatch_times_r03.json)
    python tools/dispatch_fit/emit.py && make -C fp8-mps-metal_amd && python tools/dispatch_fit/check.py

Alternating least squares on log(predicted / measured), soft-L1: per-kernel constants with the globals fixed, then the globals with the kernels
fixed (the max() terms of the model have flat regions: a few starting points per kernel)."""
import json, math, os, sys
import numpy as np
from scipy.optimize import least_squares
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from data import load_fixture, load_raw
from model import TILES, tile_predict, mx_predict, skinny_predict

data = load_raw(sys.argv[1]) if len(sys.argv) > 1 else load_fixture()
ESZ = {"bf16": 2, "f32": 4}
names = list(TILES)
rows = {n: [(M, K, N, o, d[n]) for (M, K, N, o), d in data.items() if n in d and M > 1] for n in names + ["mx", "skinny"]}
g = np.array([116000.0, 34000.0, 0.4, 0.0])
ps = {n: np.array([3.0, 0.15, 0.7, 0.2, 2.0, 0.7, 0.05, 6.0]) for n in names}
LO, HI = np.array([0, 0, 0, 0, 0, 0, 0, 2.0]), np.array([30, 5, 1, 2, 30, 20, 1.0, 9.0])
GS = np.array([1e5, 1e4, 1.0, 1.0])


def res_kernel(n, p, g):
    return [math.log(max(tile_predict(n, g, p, M, N, K, ESZ[o]), 0.1) / t) for (M, K, N, o, t) in rows[n]]


for it in range(3):
    for n in names:
        best = None
        for mf in (0.05, 0.15, 0.4):
            p0 = ps[n].copy(); p0[1] = mf
            r = least_squares(lambda p: res_kernel(n, p, g), p0, bounds=(LO, HI), loss="soft_l1", f_scale=0.1)
            if best is None or r.cost < best.cost: best = r
        ps[n] = best.x

    def res_g(gs):
        out = []
        for n in names: out += res_kernel(n, ps[n], gs * GS)
        return out
    r = least_squares(res_g, g / GS, bounds=(np.array([0.2, 0.2, 0.0, -1.0]), np.array([4.0, 10.0, 1.5, 1.0])), loss="soft_l1", f_scale=0.1)
    g = r.x * GS
    print("pass", it, "globals", np.round(g[:3], 3), "cost", round(r.cost, 3), flush=True)
consts = {"global": [float(v) for v in g[:3]], "tiles": {n: [float(v) for v in ps[n]] for n in names}}
for n in names:
    e = np.abs(np.array(res_kernel(n, ps[n], g)))
    print(f"{n:9s} n={len(rows[n]):5d} |log err| median {np.median(e):.3f} 90% {np.percentile(e, 90):.3f} max {e.max():.3f}")
for name, fn, p0 in (("mx", mx_predict, [3.5, 0.15, 0.05, 0.3, 4.0, 0.3, 0.1]), ("skinny", skinny_predict, [4, 0.15, 0.1, 0.5, 2.0, 0.3, 0.1])):
    def resid(p):
        return [math.log(max(fn(p, M, N, K, ESZ[o]), 0.1) / t) for (M, K, N, o, t) in rows[name]]
    best = None
    for s4 in (1.0, 2.0, 4.0, 8.0):
        p = list(p0); p[4] = s4
        r = least_squares(resid, np.array(p), bounds=([0, 0, 0, 0, 0.5, 0, 0], [20, 2, 2, 5, 16, 5, 5]), loss="soft_l1", f_scale=0.1)
        if best is None or r.cost < best.cost: best = r
    e = np.abs(np.array(resid(best.x)))
    print(f"{name:9s} n={len(rows[name]):5d} |log err| median {np.median(e):.3f} 90% {np.percentile(e, 90):.3f} max {e.max():.3f}")
    consts[name] = [float(v) for v in best.x]
json.dump(consts, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "constants.json"), "w"), indent=1)
print("wrote constants.json")
