#!/usr/bin/env python3
"""
Generate the golden fixtures in this directory from the REFERENCE's own
executable specification of the shader math.

Run in the build container only (the reference tree is mounted read-only at
/root/reference and does not exist on the GPU box):

    python tests/golden/make_golden.py

What it imports: /root/reference/test_fp8_correctness.py - specifically
`fp8_e4m3fn_decode_spec` (:22-50) and `fp8_e4m3fn_encode_spec` (:53-106), the
pure-Python twins of fp8_matmul.metal:19-40 and :44-92.  Nothing from the
reference is copied into the fixtures except the VALUES those functions return
and the literal test inputs the reference's tests use (cited below).

Outputs (all small, committed):
  decode_256.json      256 x {float32 bits, float16 bits} of decode_spec(b)
  encode_vectors.npz   in_bits uint32[n] -> out uint8[n] = encode_spec(float32)
  kat.json             the reference's known-answer cases and value lists
  matmul_cases.npz     seeded byte matrices + scales -> float64 products built
                       from decode_spec values (fp8_matmul.metal:116-146)
"""

import importlib.util
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("FP8_REFERENCE_DIR", "/root/reference")


def load_spec():
    path = os.path.join(REF, "test_fp8_correctness.py")
    spec = importlib.util.spec_from_file_location("ref_fp8_spec", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.fp8_e4m3fn_decode_spec, mod.fp8_e4m3fn_encode_spec


def f32_bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def bits_f32(b):
    return struct.unpack("<f", struct.pack("<I", b & 0xFFFFFFFF))[0]


def main():
    dec, enc = load_spec()

    # ---- decode: all 256 patterns --------------------------------------
    vals = [dec(b) for b in range(256)]
    f32 = np.array(vals, dtype=np.float32)
    assert all(float(a) == b for a, b in zip(f32, vals)), "decode not exact in f32"
    f16 = f32.astype(np.float16)
    assert np.array_equal(f16.astype(np.float32), f32), "decode not exact in f16"
    with open(os.path.join(HERE, "decode_256.json"), "w") as f:
        json.dump({"source": "fp8_e4m3fn_decode_spec (test_fp8_correctness.py:22-50)",
                   "f32_bits": [int(x) for x in f32.view(np.uint32)],
                   "f16_bits": [int(x) for x in f16.view(np.uint16)]}, f)

    # ---- encode vectors --------------------------------------------------
    ins = set()

    def add(x):
        b = f32_bits(float(np.float32(x)))
        ins.add(b)

    def add_around(x):
        b = f32_bits(float(np.float32(x)))
        for d in (-2, -1, 0, 1, 2):
            bb = b + d
            if 0 <= (bb & 0x7FFFFFFF) <= 0x7F800000:
                ins.add(bb & 0xFFFFFFFF)
                ins.add((bb ^ 0x80000000) & 0xFFFFFFFF)

    # every decoded value, both signs
    for v in vals:
        add_around(v)
    # every representable grid point and every rounding tie of every binade
    # from 2^-12 to 2^10 (1/16 steps cover the 1/8-step grid and its midpoints,
    # 1/32 steps the quarter points), plus +-2 ulp around each
    for e in range(-12, 11):
        for i in range(32):
            add_around((1.0 + i / 32.0) * 2.0 ** e)
    # subnormal-range grid: multiples of 2^-10 (grid + ties) up to 2^-5
    for i in range(0, 33):
        add_around(i * 2.0 ** -10)
        add_around((i + 0.5) * 2.0 ** -10)
    # saturation / boundary points
    for x in (448.0, 447.99, 449.0, 463.9, 464.0, 464.1, 479.9, 480.0, 512.0, 1e4, 3.4e38,
              float("inf"), 1e-45, 1e-40, 1e-38, 1.17549435e-38, 2.0 ** -9, 2.0 ** -10, 2.0 ** -6):
        add_around(x)
    add(0.0)
    ins.add(0x80000000)  # -0.0
    # the reference tests' own value lists (literal inputs)
    ref_lists = {
        "test_fp8_correctness.py:154-164": [0.0, 0.001953125, 0.013671875, 0.015625, 1.0, 448.0, 500.0],
        "test_fp8_metal.py:175,437": [0.0, 1.0, -1.0, 0.5, -0.5, 100.0, -100.0, 448.0],
        "test_fp8_metal.py:527-528": [1.0, 2.5, -3.0, 0.5, 10.0, -8.0, 0.0, 100.0],
        "test_fp8_metal.py:603": [0.1, 0.5, 1.0, 2.0],
        "test_fp8_metal.py:625": [10.0, 50.0, 100.0, 200.0],
        "test_fp8_metal.py:644": [0.1, 1.0, 10.0, 100.0],
        "test_fp8_metal.py:679;validate_fix.py:110": [1.0, 5.0, 10.0, 50.0],
        "test_fp8_metal.py:424": [3.14],
        "test_mps_vs_cpu.py:82-86": [0.0, 0.001, 0.01, 0.1, 0.5, 1.0, 2.0, 3.0, 10.0, 50.0, 100.0,
                                     200.0, 300.0, 400.0, 448.0, -0.001, -0.1, -1.0, -10.0, -100.0, -448.0],
        "test_mps_vs_cpu.py:219": [0.1, 0.5, 1.0, 2.0, 5.0, 10.0, 50.0, 100.0, 200.0],
        "test_mps_vs_cpu.py:303": [0.5, 1.0, 2.0, 10.0, 100.0],
        "test_cross_validation.py:52-55": [0.0, 0.1, 0.5, 1.0, 2.0, 10.0, 50.0, 100.0, 200.0, 448.0,
                                           -0.1, -1.0, -10.0, -100.0, -448.0],
        "validate_fix.py:53": [1.0, 2.0, 5.0, 10.0, 50.0, 100.0],
        "validate_fix.py:136": [0.0, 0.1, 0.5, 1.0, 10.0, 100.0, 440.0],
        "FIX_DOCUMENTATION.md:80-82": [100.0, -0.001, 0.0186],
    }
    for lst in ref_lists.values():
        for x in lst:
            add(x)
    # seeded random coverage: log-uniform magnitudes over [2^-14, 2^11) and N(0,1)*16
    rng = np.random.default_rng(20260220)
    mag = np.exp2(rng.uniform(-14.0, 11.0, size=60000)).astype(np.float32)
    sgn = np.where(rng.integers(0, 2, size=mag.size) == 1, -1.0, 1.0).astype(np.float32)
    for b in (mag * sgn).view(np.uint32):
        ins.add(int(b))
    for b in (rng.standard_normal(40000).astype(np.float32) * np.float32(16.0)).view(np.uint32):
        ins.add(int(b))
    # random raw bit patterns (finite only)
    raw = rng.integers(0, 2 ** 32, size=40000, dtype=np.uint64).astype(np.uint32)
    raw = raw[(raw & 0x7FFFFFFF) <= 0x7F800000]
    for b in raw:
        ins.add(int(b))

    in_bits = np.array(sorted(ins), dtype=np.uint32)
    out = np.empty(in_bits.size, dtype=np.uint8)
    for i, b in enumerate(in_bits):
        out[i] = enc(bits_f32(int(b)))
    np.savez_compressed(os.path.join(HERE, "encode_vectors.npz"), in_bits=in_bits, out=out)

    # ---- known answers ---------------------------------------------------
    kat = {
        "encode_known_answers": [  # test_fp8_correctness.py:154-164 (+ docs)
            [0.0, 0x00], [0.001953125, 0x01], [0.013671875, 0x07], [0.015625, 0x08],
            [1.0, 0x38], [448.0, 0x7E], [500.0, 0x7E],
            [100.0, 0x6C], [-0.001, 0x80], [0.0186, 0x0A],  # FIX_DOCUMENTATION.md:80-82
        ],
        "roundtrip_exceptions": [0x7F, 0xFF, 0x80],  # test_fp8_correctness.py:118-131
        "value_lists": {k: [[x, int(enc(float(np.float32(x))))] for x in v] for k, v in ref_lists.items()},
    }
    for x, b in kat["encode_known_answers"]:
        assert enc(float(np.float32(x))) == b, (x, b)
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1)

    # ---- matmul cases ----------------------------------------------------
    lut64 = np.array(vals, dtype=np.float64)
    cases = {}
    shapes = [(4, 64, 8), (64, 256, 128), (1, 512, 256), (33, 100, 17), (16, 128, 48), (1, 4096, 64)]
    for ci, (M, K, N) in enumerate(shapes):
        r = np.random.default_rng(1234 + ci)
        A = r.integers(0, 256, size=(M, K), dtype=np.uint8)  # includes 0x7F/0xFF on purpose
        B = r.integers(0, 256, size=(N, K), dtype=np.uint8)
        sa1 = np.array([0.01], dtype=np.float32)
        sb1 = np.array([0.02], dtype=np.float32)
        saM = r.uniform(0.005, 0.02, size=M).astype(np.float32)
        sbN = r.uniform(0.005, 0.02, size=N).astype(np.float32)
        acc = lut64[A] @ lut64[B].T  # exact products, float64 sums (metal:116-141)
        cases[f"c{ci}_A"] = A
        cases[f"c{ci}_B"] = B
        cases[f"c{ci}_sa1"] = sa1
        cases[f"c{ci}_sb1"] = sb1
        cases[f"c{ci}_saM"] = saM
        cases[f"c{ci}_sbN"] = sbN
        cases[f"c{ci}_acc"] = acc
        cases[f"c{ci}_out_tensor"] = acc * float(sa1[0]) * float(sb1[0])  # metal:144-146, mode 0
        cases[f"c{ci}_out_row"] = acc * saM.astype(np.float64)[:, None] * sbN.astype(np.float64)[None, :]
    cases["n_cases"] = np.array(len(shapes))
    np.savez_compressed(os.path.join(HERE, "matmul_cases.npz"), **cases)

    print(f"decode_256.json, encode_vectors.npz ({in_bits.size} vectors), kat.json, matmul_cases.npz written")


if __name__ == "__main__":
    sys.exit(main())
