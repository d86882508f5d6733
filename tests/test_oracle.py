"""CPU: the oracle against the golden vectors taken from the reference's own
executable spec (tests/golden/, see make_golden.py), its known-answer tests,
and torch-CPU as a second opinion.  No GPU."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch


def _load_vectors(golden_dir):
    d = np.load(os.path.join(golden_dir, "encode_vectors.npz"))
    return d["in_bits"].view(np.float32).copy(), d["out"]


def test_decode_256_bit_exact(oracle, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "decode_256.json")))
    lut = oracle.decode_lut()
    assert np.array_equal(lut.view(np.uint32), np.array(g["f32_bits"], dtype=np.uint32))
    h = oracle.dequantize_f16(np.arange(256, dtype=np.uint8), 1.0)
    assert np.array_equal(h.view(np.uint16), np.array(g["f16_bits"], dtype=np.uint16))
    # the reference's documented decode facts: NaN bytes -> +0, 0x80 -> -0
    assert lut[0x7F] == 0 and lut[0xFF] == 0 and not np.signbit(lut[0xFF])
    assert lut[0x80] == 0 and np.signbit(lut[0x80])
    assert lut[0x38] == 1.0 and lut[0x7E] == 448.0 and lut[0x01] == 2.0 ** -9 and lut[0x08] == 2.0 ** -6


def test_decode_matches_torch_except_nan(oracle):
    t = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).float().numpy()
    lut = oracle.decode_lut()
    keep = np.ones(256, bool)
    keep[[0x7F, 0xFF]] = False
    assert np.array_equal(lut[keep].view(np.uint32), t[keep].view(np.uint32))
    assert np.isnan(t[0x7F]) and np.isnan(t[0xFF])
    assert np.array_equal(oracle.decode_lut(nan_to_zero=False)[keep], lut[keep])


def test_encode_vectors(oracle, golden_dir):
    x, exp = _load_vectors(golden_dir)
    assert np.array_equal(oracle.encode(x), exp)


def test_encode_scalar_restatement(oracle, golden_dir):
    x, exp = _load_vectors(golden_dir)
    idx = np.random.default_rng(0).choice(x.size, size=20000, replace=False)
    got = np.array([oracle.encode_scalar(float(v)) for v in x[idx]], dtype=np.uint8)
    assert np.array_equal(got, exp[idx])


def test_encode_known_answers(oracle, golden_dir):
    kat = json.load(open(os.path.join(golden_dir, "kat.json")))
    for x, b in kat["encode_known_answers"]:
        assert int(oracle.encode(np.array([x], dtype=np.float32))[0]) == b, (x, b)
    for name, pairs in kat["value_lists"].items():
        xs = np.array([p[0] for p in pairs], dtype=np.float32)
        assert np.array_equal(oracle.encode(xs), np.array([p[1] for p in pairs], dtype=np.uint8)), name


def test_roundtrip_all_256(oracle, golden_dir):
    """enc(dec(b)) == b except 0x7F/0xFF/0x80 -> 0x00 (test_fp8_correctness.py:109-144)."""
    kat = json.load(open(os.path.join(golden_dir, "kat.json")))
    b = np.arange(256, dtype=np.uint8)
    rt = oracle.encode(oracle.decode(b))
    for i in range(256):
        if i in kat["roundtrip_exceptions"]:
            assert rt[i] == 0x00
        else:
            assert rt[i] == i


def test_monotone_and_error_bound(oracle):
    """Decode is monotone over 0x00..0x7E (test_fp8_correctness.py:190-222) and
    normal-range quantisation error stays under 7 % (:18, :268-288)."""
    lut = oracle.decode_lut()
    assert np.all(np.diff(lut[:0x7F]) > 0)
    vals = np.array([2.0 ** e * m for e in range(-9, 9) for m in (1.0, 1.5, 2.0, 3.0, 5.0, 7.0)], dtype=np.float32)
    vals = vals[(vals >= 0.015625) & (vals < 448.0)]
    back = oracle.decode(oracle.encode(vals))
    assert np.max(np.abs(back - vals) / vals) < 0.07


def test_encode_divergence_classes_from_torch(oracle):
    """The four documented classes where the reference encoder differs from an
    OCP/torch cast."""
    enc = lambda v: int(oracle.encode(np.array([v], dtype=np.float32))[0])
    rne = lambda v: int(oracle.encode_torch_rne(np.array([v], dtype=np.float32))[0])
    assert (enc(1.9375), rne(1.9375)) == (0x3F, 0x40)            # clamp, no carry
    assert (enc(0.0015), rne(0.0015)) == (0x00, 0x01)            # (2^-10, 2^-9) flushed
    assert (enc(470.0), rne(470.0)) == (0x7E, 0x7F)              # saturate vs NaN
    assert (enc(-0.0), rne(-0.0)) == (0x00, 0x80)                # -0.0
    assert enc(float("inf")) == 0x7E and enc(-1e30) == 0xFE


def test_encode_rne_equals_torch_cpu(oracle, golden_dir):
    x, _ = _load_vectors(golden_dir)
    t = torch.from_numpy(x.copy()).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    assert np.array_equal(oracle.encode_torch_rne(x), t)


def test_torch_cpu_five_values_agree_in_both_modes(oracle):
    """test_mps_vs_cpu.py:303 - bytes of [0.5,1,2,10,100] equal torch-CPU's."""
    v = np.array([0.5, 1.0, 2.0, 10.0, 100.0], dtype=np.float32)
    t = torch.from_numpy(v).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    assert np.array_equal(oracle.encode(v), t) and np.array_equal(oracle.encode_torch_rne(v), t)


def test_matmul_cases(oracle, golden_dir):
    d = np.load(os.path.join(golden_dir, "matmul_cases.npz"))
    for ci in range(int(d["n_cases"])):
        A, B = d[f"c{ci}_A"], d[f"c{ci}_B"]
        for sa, sb, out in ((d[f"c{ci}_sa1"], d[f"c{ci}_sb1"], d[f"c{ci}_out_tensor"]),
                            (d[f"c{ci}_saM"], d[f"c{ci}_sbN"], d[f"c{ci}_out_row"])):
            got64 = oracle.scaled_mm(A, B, sa, sb, accumulate="f64")
            assert np.allclose(got64, out, rtol=1e-12, atol=0)
            got32 = oracle.scaled_mm(A, B, sa, sb)
            bound = oracle.abs_dot_bound(A, B, sa, sb)
            assert np.all(np.abs(got32 - out) <= 2e-6 * bound + 1e-30)


def test_torch_cpu_scaled_mm_second_opinion(oracle):
    """torch-CPU _scaled_mm on NaN-free e4m3 operands equals the oracle."""
    rng = np.random.default_rng(5)
    A = rng.integers(0, 256, size=(16, 64), dtype=np.uint8)
    B = rng.integers(0, 256, size=(32, 64), dtype=np.uint8)
    A[(A & 0x7F) == 0x7F] = 0x3C
    B[(B & 0x7F) == 0x7F] = 0x3C
    ta = torch.from_numpy(A).view(torch.float8_e4m3fn)
    tb = torch.from_numpy(B).view(torch.float8_e4m3fn)
    one = torch.ones(1)
    try:
        ref = torch._scaled_mm(ta, tb.t(), scale_a=one, scale_b=one, out_dtype=torch.float32).numpy()
    except Exception as e:  # not every torch build has the CPU kernel
        pytest.skip(f"torch-CPU _scaled_mm unavailable: {e}")
    got = oracle.scaled_mm(A, B, [1.0], [1.0])
    assert np.allclose(got, ref, rtol=1e-5, atol=1e-3)


def test_rel_rmse_reproduces_reference_figure(oracle):
    """The README's 4.0 % rel-RMSE (README.md:86) at 64x256x128 with amax scaling."""
    rng = np.random.default_rng(1234)
    A = rng.standard_normal((64, 256)).astype(np.float32)
    B = rng.standard_normal((128, 256)).astype(np.float32)
    qa, ia = oracle.quantize(A)
    qb, ib = oracle.quantize(B)
    out = oracle.scaled_mm(qa, qb, [ia], [ib])
    r = oracle.rel_rmse(out, A @ B.T)
    assert 0.03 < r < 0.045, r


def test_quantize_contract(oracle):
    x = np.array([0.0, 1.0, -1.0, 0.5, -0.5, 100.0, -100.0, 448.0], dtype=np.float32)  # test_fp8_metal.py:175
    q, inv = oracle.quantize(x)
    assert inv == np.float32(1.0) and q[-1] == 0x7E
    back = oracle.dequantize_f16(q, inv).astype(np.float32)
    assert np.max(np.abs(back - x)) < 50  # the reference's own gate (test_fp8_metal.py:167-188)
    q0, inv0 = oracle.quantize(np.zeros(4, np.float32))
    assert inv0 == 1.0 and not q0.any()


def test_c_oracle_equals_numpy_oracle(oracle, oracle_c, golden_dir):
    vp = ctypes.c_void_p
    x, exp = _load_vectors(golden_dir)
    out = np.zeros(x.size, np.uint8)
    oracle_c.fp8o_encode(x.ctypes.data_as(vp), out.ctypes.data_as(vp), ctypes.c_size_t(x.size))
    assert np.array_equal(out, exp)
    lut = np.zeros(256, np.float32)
    oracle_c.fp8o_decode_lut(lut.ctypes.data_as(vp))
    assert np.array_equal(lut.view(np.uint32), oracle.decode_lut().view(np.uint32))
    h = np.zeros(256, np.uint16)
    b = np.arange(256, dtype=np.uint8)
    oracle_c.fp8o_dequant_f16_bits(b.ctypes.data_as(vp), h.ctypes.data_as(vp), ctypes.c_size_t(256))
    assert np.array_equal(h, oracle.dequantize_f16(b).view(np.uint16))
    d = np.load(os.path.join(golden_dir, "matmul_cases.npz"))
    for ci in range(int(d["n_cases"])):
        A, B = np.ascontiguousarray(d[f"c{ci}_A"]), np.ascontiguousarray(d[f"c{ci}_B"])
        sa, sb = d[f"c{ci}_saM"], d[f"c{ci}_sbN"]
        M, K = A.shape
        N = B.shape[0]
        C = np.zeros((M, N), np.float32)
        C64 = np.zeros((M, N), np.float64)
        args = (A.ctypes.data_as(vp), B.ctypes.data_as(vp))
        tail = (sa.ctypes.data_as(vp), sb.ctypes.data_as(vp), ctypes.c_size_t(M), ctypes.c_size_t(N),
                ctypes.c_size_t(K), ctypes.c_size_t(M), ctypes.c_size_t(N))
        oracle_c.fp8o_scaled_mm(*args, C.ctypes.data_as(vp), *tail)
        oracle_c.fp8o_scaled_mm_f64(*args, C64.ctypes.data_as(vp), *tail)
        out = d[f"c{ci}_out_row"]
        assert np.allclose(C64, out, rtol=1e-12, atol=0)
        bound = oracle.abs_dot_bound(A, B, sa, sb)
        assert np.all(np.abs(C - out) <= 2e-6 * bound + 1e-30)
