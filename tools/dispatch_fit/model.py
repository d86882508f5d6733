"""Python twin of fp8-mps-metal_amd/csrc/fp8mi_dispatch.h (the dispatch's cost model): used by fit.py to fit the constants and by check.py to
verify that the C++ says the same.  The K loop of a tile kernel is max(matrix pipe, global -> LDS stream); the stream's rate follows the share of its
lines that the XCD's L2 already holds (the harmonic mix of the MLP probe, profiles/r04_mlp_probe.txt)."""
import math

CUS = 256

# tile kernels: (BM, BN, K-steps per ring stage, workgroups per CU, splits K)
TILES = {
    "32x32": (32, 32, 2, 2, True), "32x64": (32, 64, 2, 1, True), "64x64": (64, 64, 2, 1, True), "64x128": (64, 128, 2, 1, True),
    "128x64": (128, 64, 2, 1, True), "128": (128, 128, 1, 2, True), "128D": (128, 128, 1, 1, True),
    "256x128W": (256, 128, 1, 1, False), "256W": (256, 256, 1, 1, False),
}


def cdiv(a, b):
    return -(-a // b)


def auto_split(tiles, K, ks, cus, has_ws):
    """resolve_split (fp8mi_gemm_epi.h) for split_k = 0 (without the workspace-size clamp: the fitted sweeps never hit it)"""
    ns = cdiv(K, 128 * ks)
    S = 1
    if has_ws and tiles <= cus // 2 and ns >= 8:
        S = min(cus // tiles, ns // 4, 16)
    if tiles > 1024:
        S = 1
    S = max(1, min(S, ns))
    if S > 1:
        per = cdiv(ns, S)
        S = cdiv(ns, per)
    return S, ns



def tile_terms(name, M, N, K, esz, cus=CUS, has_ws=True):
    BM, BN, ks, R, splits = TILES[name]
    tm, tn = cdiv(M, BM), cdiv(N, BN)
    tiles = tm * tn
    S, ns = auto_split(tiles, K, ks, cus, has_ws and splits)
    wg = tiles * S
    steps = cdiv(ns, S) * ks
    slots = cus * R
    whole, part = divmod(wg, slots)
    fill = part / slots
    vm, vn = M / (tm * BM), N / (tn * BN)                 # share of a tile's rows that exist, on average
    rows = BM * vm + BN * vn                              # valid rows a workgroup stages per K-step
    # tiles an XCD runs at one time: 1/8 of the first round, arranged gm m-tiles x gn n-tiles (tile_of_block: groups of 4 m-tiles)
    conc = max(1.0, min(wg, slots) / 8.0 / (S if S > 1 else 1))   # tiles sharing one K range on the XCD
    gm = min(4.0, tm, conc)
    gn = max(1.0, min(conc / gm, tn))
    uniq = gm * BM * (M / (tm * BM)) + gn * BN * (N / (tn * BN))
    h = max(0.0, 1.0 - uniq / (gm * gn * rows))          # L2 hit share of the stream
    busy = min(1.0, wg / cus)
    return dict(tiles=tiles, S=S, wg=wg, steps=steps, whole=whole, fill=fill, rows=rows, h=h, busy=busy, R=R, BM=BM, BN=BN, rows_all=BM + BN,
                out_mb=M * N * esz / 1e6, part_kb=BM * BN * 4 / 1024.0, percu=min(R, max(1.0, wg / cus)))


def tile_predict(name, g, p, M, N, K, esz, cus=CUS, has_ws=True):
    """g = [r_hit, r_miss, busy_exp, hbm_rate] (bytes per us per CU, exponent, bytes per us of the whole chip); p = [fixed, mfma_step, part_a, out_per_mb, xch_fixed, xch_per_64kb_slice, sync_step, operand bytes per us (millions) the kernel streams at best, the least any launch of the kernel takes]"""
    f = tile_terms(name, M, N, K, esz, cus, has_ws)
    r_miss = g[1] / max(f["busy"], 0.125) ** g[2]         # the fabric side is shared: fewer streaming CUs, more for each
    rate = 1.0 / ((1.0 - f["h"]) / r_miss + f["h"] / g[0])
    dma = f["rows"] * 128.0 * f["percu"] / rate            # co-resident workgroups share the CU's path
    step = max(p[1] * f["percu"], dma) + p[6]
    rounds = f["whole"] + (p[2] + (1.0 - p[2]) * f["fill"] if f["fill"] > 0 else 0.0)
    if f["R"] > 1:
        rounds = rounds * f["R"] / f["percu"] if f["whole"] == 0 else rounds * f["R"] / f["R"]
    loop = max(rounds * f["steps"] * step, (M * K + N * K) / (p[7] * 1e6))   # ... and nobody streams faster than the HBM delivers the operands once
    t = p[0] + loop + p[3] * f["out_mb"] / max(f["busy"], 0.25)
    if f["S"] > 1:
        t += p[4] + p[5] * f["S"] * f["part_kb"] / 64.0
    return max(t, p[8])


def streamer(c, blocks_of_x, wg, N, K, k_chain, cus=CUS):
    mb = N * K / 1e6
    stream = mb * (c[1] + c[2] * blocks_of_x)
    chain = (k_chain / 1e3) * (c[3] + c[5] * blocks_of_x) * max(1.0, wg / (cus * c[4]))
    return c[0] + max(stream, chain) + c[6] * (wg / cus) * blocks_of_x


def mx_predict(c, M, N, K, esz=2, cus=CUS):
    mxp = 2 if M <= 2 else (4 if M <= 4 else 8)
    rows = 16 if (mxp == 8 and K > 4096) else 8
    kp = 4096.0 * (1 if K <= 4096 else (2 if K <= 8192 else 4))   # the kernel's shapes hold 1, 2 or 4 wave-steps of K per wave
    return streamer(c, mxp / 8.0, cdiv(N, rows), N, K, kp, cus)


def skinny_predict(c, M, N, K, esz=2, cus=CUS):
    return streamer(c, cdiv(M, 16), cdiv(N, 16), N, K, K, cus)


def predict(consts, kernel, M, N, K, esz, cus=CUS, has_ws=True):
    if kernel == "mx":
        return mx_predict(consts["mx"], M, N, K, esz, cus)
    if kernel == "skinny":
        return skinny_predict(consts["skinny"], M, N, K, esz, cus)
    return tile_predict(kernel, consts["global"], consts["tiles"][kernel], M, N, K, esz, cus, has_ws)
