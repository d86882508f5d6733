"""GPU (MI355X): parity of the HIP path with the oracle and the golden vectors.

Every call goes through the product's op layer -> ctypes -> libfp8mi.so (the
C-ABI); the oracle is only the checker.  Bars:
  * casts: bit-exact (decode of all 256 patterns, every golden encode vector);
  * matmul: float32 accumulation, so |gpu - exact| <= MM_TOL * sum_k|a||b||sa||sb|
    against the float64 oracle (exact products, see oracle/fp8_oracle.py), and
    the reference's own accuracy gate (rel-RMSE vs fp32 matmul, ~4 %).
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# VALU paths (GEMV, generic): true float32 accumulation.  Error allowed relative
# to sum|a||b|: fp32 eps is 6e-8; a K = 16384 dot product accumulated in fp32 in
# any order stays well inside.
MM_TOL = 4e-6
# MFMA paths: the gfx950 fp8 matrix core does NOT accumulate like fp32.  Measured
# (tools/mfma_probe.hip, profiles/mfma_numerics_r01.txt; identical for the legacy
# v_mfma_f32_16x16x32_fp8_fp8 and the scaled v_mfma_scale_f32_16x16x128_f8f6f4):
# inside each group of 8 products the addends are aligned to the group's largest
# and bits more than ~13 binary places below it are truncated.  Worst case
# |err| <= 7 * 2^-13 * sum|a||b| (~8.5e-4); on random bytes the observed maximum is
# ~1e-4 and operands within a 2^12 product range are summed exactly.
MFMA_TOL = 1.0e-3       # hard bound, any input
MFMA_RMS_TOL = 1.0e-4   # rms(err) / rms(sum|a||b|), random inputs
REL_RMSE_GATE = 0.045  # reference reports 4.0 % (README.md:86), gates at 15 %

import fp8_mi355x_lib as L  # noqa: E402

# every LDS-tiled MFMA kernel of the product library (the schedule variants and the producer/consumer kernel live in the
# diagnostic library only: tools/check_kernel.py checks those against the oracle)
TILE_KERNELS = [L.KERNEL_GEMM_128, L.KERNEL_GEMM_128x64, L.KERNEL_GEMM_256, L.KERNEL_GEMM_64x128, L.KERNEL_GEMM_64x64, L.KERNEL_GEMM_32x64, L.KERNEL_GEMM_32x32,
                L.KERNEL_GEMM_128D]


def dev(x, cuda, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(cuda)


def clean_bytes(rng, shape):
    b = rng.integers(0, 256, size=shape, dtype=np.uint8)
    b[(b & 0x7F) == 0x7F] ^= 0x01  # 0x7F -> 0x7E, 0xFF -> 0xFE
    return b


def uses_mfma(kernel, M, K):
    if kernel in TILE_KERNELS or kernel in (L.KERNEL_SKINNY, L.KERNEL_GEMV_MX, L.KERNEL_GEMM_256W, L.KERNEL_GEMM_256x128W):
        return True
    if M == 1 and kernel in (L.KERNEL_AUTO, L.KERNEL_GEMV):
        return K % 16 == 0 and K > 4096   # the vec-mat hands deep K to the matrix core; K <= 4096 (config C1) stays fp32 FMA
    return kernel == L.KERNEL_AUTO and M > 1 and K % 16 == 0 and K > 0


def check_mm(oracle, native, cuda, A, B, sa, sb, *, kernel=L.KERNEL_AUTO, bias=None, scale_result=None,
             out_dtype=None, nan_mode=None, tol=None, split_k=0):
    mfma = uses_mfma(kernel, A.shape[0], A.shape[1])
    if tol is None:
        tol = MFMA_TOL if mfma else MM_TOL
    kw = {}
    if bias is not None:
        kw["bias"] = dev(bias, cuda)
    if scale_result is not None:
        kw["scale_result"] = dev(np.array([scale_result], np.float32), cuda)
    got = native.fp8_scaled_mm(dev(A, cuda), dev(B, cuda), dev(np.asarray(sa, np.float32), cuda),
                               dev(np.asarray(sb, np.float32), cuda), out_dtype=out_dtype, kernel=kernel,
                               nan_mode=nan_mode, split_k=split_k, **kw)
    torch.cuda.synchronize()
    assert got.shape == (A.shape[0], B.shape[0])
    assert got.dtype == (out_dtype or torch.float32)
    exact = oracle.scaled_mm(A, B, sa, sb, accumulate="f64")
    bound = oracle.abs_dot_bound(A, B, sa, sb)
    if bias is not None:
        exact = exact + np.asarray(bias, np.float64)[None, :]
        bound = bound + np.abs(np.asarray(bias, np.float64))[None, :]
    if scale_result is not None:
        exact = exact * scale_result
        bound = bound * abs(scale_result)
    g = got.float().cpu().numpy().astype(np.float64)
    if out_dtype in (torch.bfloat16, torch.float16):
        # one extra rounding to the 8- / 11-bit significand
        eps = 2.0 ** -8 if out_dtype == torch.bfloat16 else 2.0 ** -11
        tiny = 2.0 ** -24 if out_dtype == torch.float16 else 0.0   # fp16 subnormal quantum (results below 6e-5)
        assert np.all(np.abs(g - exact) <= tol * bound + eps * np.abs(exact) + tiny + 1e-30)
    else:
        ratio = np.max(np.abs(g - exact) / (bound + 1e-300))
        assert np.all(np.abs(g - exact) <= tol * bound + 1e-30), f"max err/bound = {ratio:.3e}"
        if mfma and tol == MFMA_TOL and g.size >= 64:
            rms = np.sqrt(np.mean((g - exact) ** 2)) / (np.sqrt(np.mean(bound ** 2)) + 1e-300)
            assert rms <= MFMA_RMS_TOL, f"rms err / rms bound = {rms:.3e}"
    return got


# ---------------------------------------------------------------------------
# casts
# ---------------------------------------------------------------------------

def test_decode_all_256_patterns_bit_exact(native, cuda, golden_dir):
    """The 256-pattern decode gate (test_fp8_metal.py:53-94, BASELINE configs[4])."""
    g = json.load(open(os.path.join(golden_dir, "decode_256.json")))
    b = torch.arange(256, dtype=torch.uint8, device=cuda)
    h = native.fp8_dequantize(b, torch.tensor([1.0]))
    assert h.dtype == torch.float16
    assert np.array_equal(h.cpu().view(torch.int16).numpy().view(np.uint16), np.array(g["f16_bits"], np.uint16))
    h2 = native.fp8_dequantize(b, None)
    assert torch.equal(h2.view(torch.int16), h.view(torch.int16))
    f = native.fp8_dequantize(b, None, out_dtype=torch.float32)
    assert np.array_equal(f.cpu().numpy().view(np.uint32), np.array(g["f32_bits"], np.uint32))
    bf = native.fp8_dequantize(b, None, out_dtype=torch.bfloat16)
    assert torch.equal(bf.float().cpu(), f.cpu())  # every e4m3 value is exact in bf16 (8-bit significand)


@pytest.mark.parametrize("count,offset", [(1, 0), (15, 0), (16, 0), (17, 0), (1000, 0), (4099, 0), (1 << 20, 0),
                                          (1000, 1), (4099, 3), (77, 16)])
def test_dequant_sizes_alignment_and_scale(native, cuda, oracle, count, offset):
    rng = np.random.default_rng(count + offset)
    raw = rng.integers(0, 256, size=count + offset, dtype=np.uint8)
    t = dev(raw, cuda)[offset:]
    for scale in (None, 0.5, 0.0137, 300.0):
        s = None if scale is None else torch.tensor([scale])
        got = native.fp8_dequantize(t, s)
        exp = oracle.dequantize_f16(raw[offset:], 1.0 if scale is None else scale)
        assert np.array_equal(got.cpu().view(torch.int16).numpy().view(np.uint16), exp.view(np.uint16)), scale
    # float32 / bfloat16 outputs = conversions of that half value (fp8_mps_patch.py:219-221)
    got32 = native.fp8_dequantize(t, torch.tensor([0.0137]), out_dtype=torch.float32)
    exp16 = torch.from_numpy(oracle.dequantize_f16(raw[offset:], 0.0137))
    assert torch.equal(got32.cpu(), exp16.float())
    gotbf = native.fp8_dequantize(t, torch.tensor([0.0137]), out_dtype=torch.bfloat16)
    assert torch.equal(gotbf.cpu(), exp16.float().to(torch.bfloat16))


def test_dequant_shapes_and_empty(native, cuda):
    e = native.fp8_dequantize(torch.empty(0, dtype=torch.uint8, device=cuda), None)
    assert e.shape == (0,) and e.dtype == torch.float16
    x = torch.randint(0, 256, (3, 5, 7), dtype=torch.uint8, device=cuda)
    assert native.fp8_dequantize(x, None).shape == (3, 5, 7)
    assert native.fp8_dequantize(x.cpu(), None).device.type == "cuda"  # moved, as the reference moves to mps


def test_encode_golden_vectors_byte_exact(native, cuda, golden_dir):
    d = np.load(os.path.join(golden_dir, "encode_vectors.npz"))
    x = d["in_bits"].view(np.float32)
    got = native.fp8_encode(dev(x, cuda))
    assert got.dtype == torch.uint8
    bad = np.nonzero(got.cpu().numpy() != d["out"])[0]
    assert bad.size == 0, [(float(x[i]), hex(int(got[i])), hex(int(d["out"][i]))) for i in bad[:5]]


def test_encode_known_answers_and_value_lists(native, cuda, golden_dir):
    kat = json.load(open(os.path.join(golden_dir, "kat.json")))
    xs = torch.tensor([p[0] for p in kat["encode_known_answers"]], dtype=torch.float32, device=cuda)
    assert native.fp8_encode(xs).cpu().tolist() == [p[1] for p in kat["encode_known_answers"]]
    for name, pairs in kat["value_lists"].items():
        xs = torch.tensor([p[0] for p in pairs], dtype=torch.float32, device=cuda)
        assert native.fp8_encode(xs).cpu().tolist() == [p[1] for p in pairs], name


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_encode_half_inputs(native, cuda, oracle, dtype):
    rng = np.random.default_rng(7)
    x = torch.from_numpy((rng.standard_normal(100003) * 16).astype(np.float32)).to(dtype)
    special = torch.tensor([0.0, -0.0, 448.0, 449.0, 1e4, -1e4, 6e-5, 2.0 ** -9, 2.0 ** -10, 0.0019, float("inf")]).to(dtype)
    x = torch.cat([x, special])
    got = native.fp8_encode(x.to(cuda)).cpu().numpy()
    assert np.array_equal(got, oracle.encode(x.float().numpy()))


@pytest.mark.parametrize("count,offset", [(1, 0), (15, 0), (17, 0), (4099, 0), (1000, 1), (4099, 3)])
def test_encode_sizes_and_alignment(native, cuda, oracle, count, offset):
    rng = np.random.default_rng(100 + count + offset)
    raw = (rng.standard_normal(count + offset) * 30).astype(np.float32)
    got = native.fp8_encode(dev(raw, cuda)[offset:])
    assert np.array_equal(got.cpu().numpy(), oracle.encode(raw[offset:]))


def test_encode_rne_mode_equals_torch_cpu(native, cuda, golden_dir):
    d = np.load(os.path.join(golden_dir, "encode_vectors.npz"))
    x = torch.from_numpy(d["in_bits"].view(np.float32).copy())
    x = torch.cat([x, torch.tensor([float("nan"), -float("nan")])])
    exp = x.to(torch.float8_e4m3fn).view(torch.uint8)
    got = native.fp8_encode(x.to(cuda), encode_mode=L.ENC_RNE).cpu()
    assert torch.equal(got[:-2], exp[:-2])
    assert (got[-2:] & 0x7F).tolist() == [0x7F, 0x7F]
    # the vector kernel runs the hardware convert (v_cvt_pk_fp8_f32) with explicit edges, the scalar kernel (misaligned
    # views) the all-integer form: both must give torch-CPU's bytes, for every source dtype (test_mps_vs_cpu.py:283-357)
    xd = x[:-2].to(cuda)
    assert torch.equal(native.fp8_encode(xd[1:], encode_mode=L.ENC_RNE).cpu(), exp[1:-2])
    edges = torch.tensor([2.0 ** -10, -(2.0 ** -10), 2.0 ** -10 * (1 + 2.0 ** -23), 2.0 ** -9 * 0.75, 2.0 ** -9, 1.5 * 2.0 ** -9,
                          448.0, 463.99997, 464.0, 464.00003, -464.00003, 480.0, 1e30, float("inf"), -float("inf"), 0.0, -0.0,
                          1.9375, 0.0146484375, 2.0 ** -6 * (1 - 2.0 ** -5)] * 16)
    assert torch.equal(native.fp8_encode(edges.to(cuda), encode_mode=L.ENC_RNE).cpu(), edges.to(torch.float8_e4m3fn).view(torch.uint8))
    for dt in (torch.float16, torch.bfloat16):
        xs = x[:-2].to(dt)
        xs = xs[torch.isfinite(xs)]
        e2 = xs.to(torch.float8_e4m3fn).view(torch.uint8)
        assert torch.equal(native.fp8_encode(xs.to(cuda), encode_mode=L.ENC_RNE).cpu(), e2), dt
        assert torch.equal(native.fp8_encode(xs.to(cuda)[3:], encode_mode=L.ENC_RNE).cpu(), e2[3:]), dt
    # amax-scaled quantisation in this mode: bytes = torch-CPU cast of x * float32(448 / amax)
    g = torch.Generator().manual_seed(4)
    y = torch.randn(70001, generator=g) * 5
    q, inv = native.fp8_quantize(y.to(cuda), encode_mode=L.ENC_RNE)
    scale = np.float32(448.0 / float(y.abs().max()))
    assert torch.equal(q.cpu(), (y * float(scale)).to(torch.float8_e4m3fn).view(torch.uint8))


def test_encode_other_dtypes_and_shapes(native, cuda, oracle):
    assert native.fp8_encode(torch.empty(0, device=cuda)).shape == (0,)
    i = torch.arange(-20, 20, dtype=torch.int64, device=cuda).reshape(4, 10)
    got = native.fp8_encode(i)
    assert got.shape == (4, 10)
    assert np.array_equal(got.cpu().numpy().ravel(), oracle.encode(np.arange(-20, 20, dtype=np.float32)))
    d = torch.tensor([3.14], dtype=torch.float64)
    assert native.fp8_encode(d).cpu().tolist() == oracle.encode(np.array([3.14], np.float32)).tolist()


def test_amax_and_quantize(native, cuda, oracle):
    rng = np.random.default_rng(11)
    for n in (1, 7, 1000, 262147):
        x = (rng.standard_normal(n) * 3).astype(np.float32)
        assert float(native.fp8_amax(dev(x, cuda)).cpu()) == float(np.max(np.abs(x)))
        q, inv = native.fp8_quantize(dev(x, cuda))
        eq, einv = oracle.quantize(x)
        assert inv.shape == (1,) and inv.dtype == torch.float32
        assert float(inv.cpu()) == float(einv)
        assert np.array_equal(q.cpu().numpy(), eq)
    # bf16 source, zero input, quantize -> dequantize gate of test_fp8_metal.py:167-188
    xb = torch.from_numpy((rng.standard_normal(4099) * 3).astype(np.float32)).to(torch.bfloat16)
    q, inv = native.fp8_quantize(xb.to(cuda))
    eq, einv = oracle.quantize(xb.float().numpy())
    assert np.array_equal(q.cpu().numpy(), eq) and float(inv.cpu()) == float(einv)
    q0, inv0 = native.fp8_quantize(torch.zeros(33, device=cuda))
    assert float(inv0.cpu()) == 1.0 and not q0.any()
    qe, inve = native.fp8_quantize(torch.empty(0, device=cuda))
    assert qe.numel() == 0 and float(inve.cpu()) == 1.0
    x = torch.tensor([0.0, 1.0, -1.0, 0.5, -0.5, 100.0, -100.0, 448.0], device=cuda)
    q, inv = native.fp8_quantize(x)
    back = native.fp8_dequantize(q, inv).float()
    assert float((back - x).abs().max()) < 50


def test_cast_roundtrip_property_large(native, cuda, oracle):
    """2^27 elements: enc(dec(enc(x))) == enc(x) everywhere (idempotence, modulo -0), and
    a 2^20-element sample equals the oracle byte for byte."""
    n = 1 << 27
    g = torch.Generator(device=cuda).manual_seed(1234)
    x = torch.randn(n, device=cuda, generator=g) * 16
    q = native.fp8_encode(x)
    h = native.fp8_dequantize(q, None)
    q2 = native.fp8_encode(h)
    # 0x80 (what small negatives flush to) decodes to -0.0, which re-encodes to 0x00:
    # the reference's own round-trip exception (test_fp8_correctness.py:118-131)
    assert torch.equal(torch.where(q == 0x80, torch.zeros_like(q), q), q2)
    assert bool((q == 0x80).any())
    idx = torch.randint(0, n, (1 << 20,), device=cuda, generator=g)
    assert np.array_equal(q[idx].cpu().numpy(), oracle.encode(x[idx].cpu().numpy()))
    assert np.array_equal(h[idx].cpu().view(torch.int16).numpy().view(np.uint16),
                          oracle.dequantize_f16(q[idx].cpu().numpy()).view(np.uint16))
    assert not bool(((q & 0x7F) == 0x7F).any())  # the reference encoder never emits a NaN pattern


# ---------------------------------------------------------------------------
# scaled matmul: golden cases through every kernel that can run them
# ---------------------------------------------------------------------------

def _golden_cases(golden_dir):
    d = np.load(os.path.join(golden_dir, "matmul_cases.npz"))
    for ci in range(int(d["n_cases"])):
        yield ci, d[f"c{ci}_A"], d[f"c{ci}_B"], d


def _kernels_for(M, K):
    ks = [L.KERNEL_AUTO, L.KERNEL_GENERIC]
    if K % 16 == 0:
        ks += TILE_KERNELS
        if M == 1:
            ks += [L.KERNEL_GEMV, L.KERNEL_GEMV_FP32]
        if M <= 64:
            ks.append(L.KERNEL_SKINNY)
        if 2 <= M <= 8 and K <= 16384:
            ks.append(L.KERNEL_GEMV_MX)
    return ks


def test_golden_matmul_cases_all_kernels(native, cuda, oracle, golden_dir):
    """Seeded byte matrices INCLUDING NaN patterns (0x7F/0xFF decode to 0,
    fp8_matmul.metal:21), per-tensor and per-row scales."""
    for ci, A, B, d in _golden_cases(golden_dir):
        M, K = A.shape
        for kernel in _kernels_for(M, K):
            tol = MFMA_TOL if uses_mfma(kernel, M, K) else MM_TOL
            got = check_mm(oracle, native, cuda, A, B, d[f"c{ci}_sa1"], d[f"c{ci}_sb1"], kernel=kernel)
            bound = oracle.abs_dot_bound(A, B, d[f"c{ci}_sa1"], d[f"c{ci}_sb1"])
            assert np.all(np.abs(got.cpu().numpy() - d[f"c{ci}_out_tensor"]) <= tol * bound + 1e-30), (ci, kernel)
            got = check_mm(oracle, native, cuda, A, B, d[f"c{ci}_saM"], d[f"c{ci}_sbN"], kernel=kernel)
            bound = oracle.abs_dot_bound(A, B, d[f"c{ci}_saM"], d[f"c{ci}_sbN"])
            assert np.all(np.abs(got.cpu().numpy() - d[f"c{ci}_out_row"]) <= tol * bound + 1e-30), (ci, kernel)


def test_nan_modes(native, cuda, oracle):
    rng = np.random.default_rng(21)
    A = clean_bytes(rng, (32, 256))
    B = clean_bytes(rng, (48, 256))
    A[3, 17] = 0x7F
    B[5, 100] = 0xFF
    for kernel in (L.KERNEL_GEMM_128, L.KERNEL_GENERIC):
        check_mm(oracle, native, cuda, A, B, [1.0], [1.0], kernel=kernel)  # reference mode: NaN byte = 0
        got = native.fp8_scaled_mm(dev(A, cuda), dev(B, cuda), torch.ones(1), torch.ones(1), kernel=kernel,
                                   nan_mode=L.NAN_PROPAGATE).cpu().numpy()
        nan = np.isnan(got)
        assert nan[3, :].all() and nan[:, 5].all() and nan.sum() == 48 + 32 - 1  # OCP: row 3 and column 5 poisoned
    x = clean_bytes(rng, (1, 2048))
    W = clean_bytes(rng, (64, 2048))
    W[7, 5] = 0x7F
    W[9, 2047] = 0xFF
    x[0, 33] = 0xFF
    check_mm(oracle, native, cuda, x, W, [0.5], [0.25], kernel=L.KERNEL_GEMV)
    got = native.fp8_scaled_mm(dev(x, cuda), dev(W, cuda), torch.ones(1), torch.ones(1), kernel=L.KERNEL_GEMV,
                               nan_mode=L.NAN_PROPAGATE).cpu().numpy()
    assert np.isnan(got).all()  # a NaN in x poisons every output


@pytest.mark.parametrize("M,K,N", [(1, 16, 1), (1, 1040, 33), (1, 4096, 4096), (1, 14336, 4096), (1, 14352, 100),
                                   (1, 20480, 72), (1, 1024, 8), (1, 2064, 17)])
def test_gemv_shapes(native, cuda, oracle, M, K, N):
    """Vec-mat path (fp8_matmul.metal:155-210): BASELINE configs[0] (K=N=4096) and
    configs[1] (K=14336, N=4096) plus ragged K / N."""
    rng = np.random.default_rng(K + N)
    x = clean_bytes(rng, (1, K))
    W = clean_bytes(rng, (N, K))
    check_mm(oracle, native, cuda, x, W, [0.01], [0.01], kernel=L.KERNEL_GEMV)
    sw = rng.uniform(0.005, 0.02, size=N).astype(np.float32)
    check_mm(oracle, native, cuda, x, W, [0.013], sw, kernel=L.KERNEL_GEMV)
    check_mm(oracle, native, cuda, x, W, [0.013], sw)  # auto dispatch picks the same path
    # the IEEE-fp32 form of the vec-mat (the reference's accumulation, fp8_matmul.metal:177-199) at every K
    check_mm(oracle, native, cuda, x, W, [0.013], sw, kernel=L.KERNEL_GEMV_FP32, tol=MM_TOL)


def test_gemv_matrix_core_form_is_exact_on_narrow_range(native, cuda, oracle):
    """Deep-K vec-mat on the matrix core (diagonal of the 16x16 product tile): the hardware tolerance must not hide a
    software error - operands within a 2^12 product range are summed exactly, so the result meets the fp32 bound; NaN
    bytes in W / x follow both NaN modes; bias / per-row scale / bf16 epilogue as everywhere."""
    rng = np.random.default_rng(77)
    for (K, N) in ((6144, 300), (14336, 129), (20480, 33)):
        x = (0x28 + rng.integers(0, 0x20, size=(1, K))).astype(np.uint8)
        W = (0x28 + rng.integers(0, 0x20, size=(N, K))).astype(np.uint8)
        check_mm(oracle, native, cuda, x, W, [0.5], [2.0], kernel=L.KERNEL_GEMV, tol=MM_TOL)
        a = native.fp8_scaled_mm(dev(x, cuda), dev(W, cuda), torch.ones(1), torch.ones(1), kernel=L.KERNEL_GEMV)
        b = native.fp8_scaled_mm(dev(x, cuda), dev(W, cuda), torch.ones(1), torch.ones(1), kernel=L.KERNEL_GEMV_FP32)
        assert bool(((a - b).abs() <= MM_TOL * a.abs()).all())
    x = clean_bytes(rng, (1, 8192))
    W = clean_bytes(rng, (70, 8192))
    W[7, 5] = 0x7F
    W[9, 8191] = 0xFF
    sw = rng.uniform(0.005, 0.02, size=70).astype(np.float32)
    bias = rng.standard_normal(70).astype(np.float32)
    check_mm(oracle, native, cuda, x, W, [0.5], sw, kernel=L.KERNEL_GEMV, bias=bias, scale_result=0.25, out_dtype=torch.bfloat16)
    got = native.fp8_scaled_mm(dev(x, cuda), dev(W, cuda), torch.ones(1), torch.ones(1), kernel=L.KERNEL_GEMV,
                               nan_mode=L.NAN_PROPAGATE).cpu().numpy()
    assert np.isnan(got[0, 7]) and np.isnan(got[0, 9]) and np.isnan(got).sum() == 2    # OCP: only the rows holding NaN bytes
    x[0, 4099] = 0xFF
    check_mm(oracle, native, cuda, x, W, [0.5], sw, kernel=L.KERNEL_GEMV)               # reference: NaN byte = 0.0
    got = native.fp8_scaled_mm(dev(x, cuda), dev(W, cuda), torch.ones(1), torch.ones(1), kernel=L.KERNEL_GEMV,
                               nan_mode=L.NAN_PROPAGATE).cpu().numpy()
    assert np.isnan(got).all()  # a NaN in x poisons every output


@pytest.mark.parametrize("M", [2, 3, 4, 5, 7, 8])
def test_few_rows_kernel(native, cuda, oracle, M):
    """2..8 activation rows on the vec-mat's weight-streaming structure (FP8MI_KERNEL_GEMV_MX; the reference runs these
    through fp8_scaled_matmul_kernel, fp8_matmul.metal:99-147): every launch shape (K <= 4096 / 8192 / 16384 per row
    count), ragged N, NaN bytes in both operands under both NaN modes, per-row scales, bias, scale_result, bf16."""
    for (K, N) in ((16, 1), (272, 70), (1040, 33), (4096, 513), (8192, 64), (14336, 257), (16384, 40)):
        rng = np.random.default_rng(M * 1000 + K + N)
        X = clean_bytes(rng, (M, K))
        W = clean_bytes(rng, (N, K))
        sa = rng.uniform(0.005, 0.02, size=M).astype(np.float32)
        sb = rng.uniform(0.005, 0.02, size=N).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        a = check_mm(oracle, native, cuda, X, W, [0.01], [0.02], kernel=L.KERNEL_GEMV_MX)
        if K > 16:
            b = check_mm(oracle, native, cuda, X, W, [0.01], [0.02])   # what AUTO picks agrees with the oracle too ...
            if L.load().fp8mi_choose_kernel(M, N, K, K, K, N, L.F32, 1, 0) == L.KERNEL_GEMV_MX:
                assert torch.equal(a, b)                                # ... and, where that is this kernel, bit for bit
        check_mm(oracle, native, cuda, X, W, sa, sb, kernel=L.KERNEL_GEMV_MX, bias=bias, scale_result=0.25)
        check_mm(oracle, native, cuda, X, W, sa, sb, kernel=L.KERNEL_GEMV_MX, bias=bias, out_dtype=torch.bfloat16)
        X[M - 1, K // 2] = 0x7F
        W[N - 1, K - 1] = 0xFF
        W[0, 0] = 0x7F
        check_mm(oracle, native, cuda, X, W, sa, sb, kernel=L.KERNEL_GEMV_MX, bias=bias)     # reference: NaN byte = 0.0
        got = native.fp8_scaled_mm(dev(X, cuda), dev(W, cuda), torch.ones(1), torch.ones(1), kernel=L.KERNEL_GEMV_MX,
                                   nan_mode=L.NAN_PROPAGATE).cpu().numpy()
        nan = np.isnan(got)
        expect = np.zeros((M, N), bool)
        expect[M - 1, :] = True
        expect[:, N - 1] = True
        expect[:, 0] = True
        assert np.array_equal(nan, expect)                                                    # OCP: exactly the poisoned row / columns
    with pytest.raises(RuntimeError):   # outside its envelope the explicit id refuses (AUTO falls through to the other kernels)
        native.fp8_scaled_mm(torch.zeros(9, 64, dtype=torch.uint8, device=cuda), torch.zeros(8, 64, dtype=torch.uint8, device=cuda),
                             torch.ones(1), torch.ones(1), kernel=L.KERNEL_GEMV_MX)
    with pytest.raises(RuntimeError):
        native.fp8_scaled_mm(torch.zeros(4, 16400, dtype=torch.uint8, device=cuda), torch.zeros(8, 16400, dtype=torch.uint8, device=cuda),
                             torch.ones(1), torch.ones(1), kernel=L.KERNEL_GEMV_MX)


def test_few_rows_kernel_is_exact_on_narrow_range_and_matches_rows(native, cuda, oracle):
    """As for the vec-mat: operands within a 2^12 product range are summed exactly by the matrix core, so the few-rows
    kernel must meet the fp32 bound there (a wrong lane / register in the diagonal extraction cannot hide behind the
    hardware tolerance), and row m of its result equals the M = 1 kernel's result for x[m] bit for bit at deep K."""
    rng = np.random.default_rng(78)
    for (M, K, N) in ((2, 6144, 300), (5, 6144, 300), (4, 14336, 129), (8, 16384, 33), (8, 4096, 77), (3, 2048, 130)):
        X = (0x28 + rng.integers(0, 0x20, size=(M, K))).astype(np.uint8)
        W = (0x28 + rng.integers(0, 0x20, size=(N, K))).astype(np.uint8)
        check_mm(oracle, native, cuda, X, W, [0.5], [2.0], kernel=L.KERNEL_GEMV_MX, tol=MM_TOL)
    # transposed epilogue (the sharded linear's form): bias per output ROW of this call, scales in the swapped order
    M, K, N = 6, 8192, 48
    X = clean_bytes(rng, (M, K))
    W = clean_bytes(rng, (N, K))
    sx = rng.uniform(0.005, 0.02, size=M).astype(np.float32)
    bias_m = rng.standard_normal(M).astype(np.float32)
    t = native.fp8_scaled_mm(dev(X, cuda), dev(W, cuda), dev(sx, cuda), torch.full((1,), 0.03), bias=dev(bias_m, cuda),
                             kernel=L.KERNEL_GEMV_MX, transposed_epilogue=True).cpu().numpy()
    exact = oracle.scaled_mm(X, W, sx, [0.03], accumulate="f64") + bias_m[:, None]
    bound = oracle.abs_dot_bound(X, W, sx, [0.03]) + np.abs(bias_m)[:, None]
    assert np.all(np.abs(t - exact) <= MFMA_TOL * bound + 1e-30)


@pytest.mark.parametrize("kernel", TILE_KERNELS)
@pytest.mark.parametrize("M,K,N", [(1, 128, 1), (5, 16, 3), (128, 128, 128), (130, 272, 70), (300, 1040, 200),
                                   (256, 512, 256), (257, 384, 513), (64, 4096, 96),
                                   (1300, 144, 900), (700, 32, 1100)])
def test_gemm_tile_kernels_ragged(native, cuda, oracle, kernel, M, K, N):
    """MFMA tile kernels on full and ragged tiles, K tails (K % 128 != 0) included; the last two shapes
    give m-tile counts that are not multiples of the tile map's group of 4 and grids that are not
    multiples of the 8 XCDs (every output element checked, so a tile mapped twice or never shows)."""
    rng = np.random.default_rng(M * 7 + K * 3 + N)
    A = clean_bytes(rng, (M, K))
    B = clean_bytes(rng, (N, K))
    sa = rng.uniform(0.005, 0.02, size=M).astype(np.float32)
    sb = rng.uniform(0.005, 0.02, size=N).astype(np.float32)
    check_mm(oracle, native, cuda, A, B, sa, sb, kernel=kernel)
    check_mm(oracle, native, cuda, A, B, [0.01], sb, kernel=kernel)  # mixed per-tensor / per-row broadcasts


@pytest.mark.parametrize("M,K,N", [(1, 128, 16), (2, 16, 1), (4, 4096, 4096), (4, 1040, 100), (16, 4096, 512), (17, 528, 33),
                                   (4, 8192, 48), (3, 2048, 40), (40, 8192, 24),   # 8 / 2 / 4 waves per fragment
                                   (33, 2048, 200), (64, 14336, 256), (48, 144, 17)])
def test_skinny_kernel(native, cuda, oracle, M, K, N):
    """2 <= M <= 64 weight-streaming MFMA kernel (the reference's M <= 16 route,
    fp8_mps_native.py:208; M=4, K=N=4096 is one of its published shapes, README.md:77),
    NaN bytes included (scrubbed in-register)."""
    rng = np.random.default_rng(M * 11 + K + N)
    A = rng.integers(0, 256, size=(M, K), dtype=np.uint8)
    B = rng.integers(0, 256, size=(N, K), dtype=np.uint8)
    sa = rng.uniform(0.005, 0.02, size=M).astype(np.float32)
    sb = rng.uniform(0.005, 0.02, size=N).astype(np.float32)
    check_mm(oracle, native, cuda, A, B, sa, sb, kernel=L.KERNEL_SKINNY)
    bias = rng.standard_normal(N).astype(np.float32)
    check_mm(oracle, native, cuda, A, B, [0.01], sb, kernel=L.KERNEL_SKINNY, bias=bias, scale_result=0.5, out_dtype=torch.bfloat16)
    if M >= 2:
        check_mm(oracle, native, cuda, A, B, sa, [0.02])  # auto dispatch lands here for 2 <= M <= 32


def test_gemm_mfma_operand_map_asymmetric(native, cuda, oracle):
    """A = 'identity-like' selector with an ASYMMETRIC B: catches a swapped
    row/column map or a k-permutation mismatch between the two MFMA operands."""
    K = 256
    A = np.zeros((128, K), np.uint8)
    for m in range(128):
        A[m, (m * 5 + 3) % K] = 0x38  # 1.0 at a row-dependent k
    rng = np.random.default_rng(3)
    B = clean_bytes(rng, (128, K))
    got = check_mm(oracle, native, cuda, A, B, [1.0], [1.0], kernel=L.KERNEL_GEMM_128, tol=0.0)
    exp = oracle.decode(B)[:, [(m * 5 + 3) % K for m in range(128)]].T
    assert np.array_equal(got.cpu().numpy(), exp)


@pytest.mark.parametrize("kernel", TILE_KERNELS)
def test_gemm_fp32_exact_on_narrow_range(native, cuda, oracle, kernel):
    """Operands with |x| in [0.25, 4): all products lie within 2^8 of each other,
    above the matrix core's alignment cut-off, so the MFMA kernels must agree
    with the oracle to float32 rounding - a software bug cannot hide behind the
    hardware tolerance."""
    rng = np.random.default_rng(77)
    M, K, N = 200, 1024, 328
    A = (0x28 + rng.integers(0, 0x20, size=(M, K))).astype(np.uint8) | (rng.integers(0, 2, size=(M, K)).astype(np.uint8) << 7)
    B = (0x28 + rng.integers(0, 0x20, size=(N, K))).astype(np.uint8) | (rng.integers(0, 2, size=(N, K)).astype(np.uint8) << 7)
    check_mm(oracle, native, cuda, A, B, [0.5], [2.0], kernel=kernel, tol=MM_TOL)


def test_gemm_full_size_c3_against_oracle(native, cuda, oracle):
    """BASELINE configs[2]: M=512, K=N=4096 against the float64 oracle."""
    rng = np.random.default_rng(1234)
    A = clean_bytes(rng, (512, 4096))
    B = clean_bytes(rng, (4096, 4096))
    for kernel in [L.KERNEL_AUTO] + TILE_KERNELS:
        check_mm(oracle, native, cuda, A, B, [0.01], [0.01], kernel=kernel)


def test_generic_kernel_unaligned(native, cuda, oracle):
    rng = np.random.default_rng(9)
    for (M, K, N) in [(33, 100, 17), (1, 7, 5), (3, 1, 2), (2, 1023, 65)]:
        A = rng.integers(0, 256, size=(M, K), dtype=np.uint8)
        B = rng.integers(0, 256, size=(N, K), dtype=np.uint8)
        check_mm(oracle, native, cuda, A, B, [0.02], [0.5])  # auto -> generic (K % 16 != 0)
    # unaligned views of a larger buffer (row stride != K, odd base address)
    big = rng.integers(0, 256, size=(40, 300), dtype=np.uint8)
    tA = dev(big, cuda)[1:9, 3:259]      # (8, 256), stride 300, base + 303
    tB = dev(big, cuda)[10:40, 5:261]    # (30, 256)
    got = native.fp8_scaled_mm(tA, tB, torch.ones(1), torch.ones(1)).cpu().numpy()
    exact = oracle.scaled_mm(big[1:9, 3:259], big[10:40, 5:261], [1.0], [1.0], accumulate="f64")
    bound = oracle.abs_dot_bound(big[1:9, 3:259], big[10:40, 5:261], [1.0], [1.0])
    assert np.all(np.abs(got - exact) <= MM_TOL * bound + 1e-30)


@pytest.mark.parametrize("M,K,N", [(300, 4100, 272), (520, 1000, 136), (2, 4100, 4096), (4, 4099, 2048), (64, 2050, 1024)])
def test_unaligned_operands_take_the_padded_mfma_path(native, cuda, oracle, M, K, N):
    """K % 16 != 0 (the reference accepts any contiguous shape, fp8_mps_native.py:55-60): above PAD_MIN_MACS multiply-adds the op layer copies both
    operands into aligned buffers whose rows are zero-padded to the next multiple of 16 and the MFMA / vec-mat kernels run - same sum (a zero byte is
    +0.0), bit-identical to the call on explicitly padded operands, and within the matrix core's bound of the f64 oracle on the UNPADDED bytes."""
    assert M * N * K >= native.PAD_MIN_MACS
    rng = np.random.default_rng(M + K + N)
    A, B = clean_bytes(rng, (M, K)), clean_bytes(rng, (N, K))
    A[0, :] = 0xFF if M > 1 else A[0, :]     # NaN bytes (reference: decode to 0) ride along
    sa, sb = [0.02], rng.uniform(0.01, 0.03, size=N).astype(np.float32)
    got = native.fp8_scaled_mm(dev(A, cuda), dev(B, cuda), dev(np.asarray(sa, np.float32), cuda), dev(sb, cuda))
    Kp = (K + 15) // 16 * 16
    Ap, Bp = np.zeros((M, Kp), np.uint8), np.zeros((N, Kp), np.uint8)
    Ap[:, :K], Bp[:, :K] = A, B
    assert L.load().fp8mi_choose_kernel(M, N, Kp, Kp, Kp, N, L.F32, 1, 0) != L.KERNEL_GENERIC
    want = native.fp8_scaled_mm(dev(Ap, cuda), dev(Bp, cuda), dev(np.asarray(sa, np.float32), cuda), dev(sb, cuda))
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    exact = oracle.scaled_mm(A, B, sa, sb, accumulate="f64")
    bound = oracle.abs_dot_bound(A, B, sa, sb)
    assert np.all(np.abs(got.cpu().numpy().astype(np.float64) - exact) <= MFMA_TOL * bound + 1e-30)


def test_misaligned_views_take_the_padded_mfma_path(native, cuda, oracle):
    """A sliced weight view (base pointer off by 8 bytes, row stride not a multiple of 16): copied once per call into aligned rows; the forced
    generic kernel on the same views is the checker of last resort (exact fp32 sums) and the oracle the judge."""
    rng = np.random.default_rng(77)
    big = clean_bytes(rng, (600, 1100))
    tA, tB = dev(big, cuda)[8:264, 8:1032], dev(big, cuda)[300:556, 24:1048]   # (256, 1024) each: K aligned, bases and strides are not
    assert tA.data_ptr() % 16 != 0 and tA.stride(0) % 16 != 0 and 256 * 256 * 1024 >= native.PAD_MIN_MACS
    one = torch.ones(1)
    got = native.fp8_scaled_mm(tA, tB, one, one)
    slow = native.fp8_scaled_mm(tA, tB, one, one, kernel=L.KERNEL_GENERIC)
    torch.cuda.synchronize()
    exact = oracle.scaled_mm(big[8:264, 8:1032], big[300:556, 24:1048], [1.0], [1.0], accumulate="f64")
    bound = oracle.abs_dot_bound(big[8:264, 8:1032], big[300:556, 24:1048], [1.0], [1.0])
    assert np.all(np.abs(got.cpu().numpy() - exact) <= MFMA_TOL * bound + 1e-30)
    assert np.all(np.abs(slow.cpu().numpy() - exact) <= MM_TOL * bound + 1e-30)


def test_tall_problem_grid_math(native, cuda, oracle):
    """M beyond 65535 (grid.y/z split of the generic kernel, many m-tiles of the tile kernels)."""
    rng = np.random.default_rng(12)
    M = 70003
    for (K, N) in ((20, 3), (32, 24)):   # K % 16 != 0 -> generic; aligned -> MFMA tiles
        A = clean_bytes(rng, (M, K))
        B = clean_bytes(rng, (N, K))
        got = native.fp8_scaled_mm(dev(A, cuda), dev(B, cuda), torch.ones(1), torch.ones(1)).cpu().numpy()
        exact = oracle.scaled_mm(A, B, [1.0], [1.0], accumulate="f64")
        assert np.all(np.abs(got - exact) <= MFMA_TOL * oracle.abs_dot_bound(A, B, [1.0], [1.0]) + 1e-30)


def test_padded_row_stride_no_copy(native, cuda, oracle):
    """lda / ldb > K (16-byte aligned): handled by the tuned kernels without a copy."""
    rng = np.random.default_rng(10)
    bufA = clean_bytes(rng, (64, 512))
    bufB = clean_bytes(rng, (96, 512))
    tA, tB = dev(bufA, cuda)[:, :256], dev(bufB, cuda)[:, :256]
    got = native.fp8_scaled_mm(tA, tB, torch.ones(1), torch.ones(1)).cpu().numpy()
    exact = oracle.scaled_mm(bufA[:, :256], bufB[:, :256], [1.0], [1.0], accumulate="f64")
    assert np.all(np.abs(got - exact) <= MFMA_TOL * oracle.abs_dot_bound(bufA[:, :256], bufB[:, :256], [1.0], [1.0]) + 1e-30)
    x = dev(bufA, cuda)[3:4, :256]
    got = native.fp8_scaled_mm(x, tB, torch.ones(1), torch.ones(1)).cpu().numpy()
    exact = oracle.scaled_mm(bufA[3:4, :256], bufB[:, :256], [1.0], [1.0], accumulate="f64")
    assert np.all(np.abs(got - exact) <= MM_TOL * oracle.abs_dot_bound(bufA[3:4, :256], bufB[:, :256], [1.0], [1.0]) + 1e-30)


@pytest.mark.parametrize("out_dtype", [None, torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 48])
def test_fused_epilogue(native, cuda, oracle, out_dtype, M):
    """+ bias, * scale_result, cast - in the reference's order (fp8_mps_patch.py:94-104)."""
    rng = np.random.default_rng(31 + M)
    K, N = 512, 136
    A = clean_bytes(rng, (M, K))
    B = clean_bytes(rng, (N, K))
    bias = rng.standard_normal(N).astype(np.float32)
    check_mm(oracle, native, cuda, A, B, [0.01], [0.02], bias=bias, scale_result=0.5, out_dtype=out_dtype)
    check_mm(oracle, native, cuda, A, B, [0.01], [0.02], bias=bias, out_dtype=out_dtype)
    # bias in bf16, as a bf16 model would pass it
    bb = torch.from_numpy(bias).to(torch.bfloat16)
    got = native.fp8_scaled_mm(dev(A, cuda), dev(B, cuda), torch.tensor([0.01]), torch.tensor([0.02]),
                               bias=bb.to(cuda), out_dtype=torch.float32).cpu().numpy()
    exact = oracle.scaled_mm(A, B, [0.01], [0.02], accumulate="f64") + bb.float().numpy()[None, :]
    bound = oracle.abs_dot_bound(A, B, [0.01], [0.02]) + np.abs(bb.float().numpy())[None, :]
    assert np.all(np.abs(got - exact) <= (MFMA_TOL if M > 1 else MM_TOL) * bound + 1e-30)


# ---------------------------------------------------------------------------
# 256x256 tile, one wave per SIMD, hand-scheduled K loop (FP8MI_KERNEL_GEMM_256W)
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("M,K,N", [(256, 256, 256), (256, 384, 512), (512, 1024, 256), (768, 640, 1024), (256, 4096, 256),
                                   (300, 512, 256), (257, 384, 512), (129, 1024, 256), (1000, 640, 768), (1, 256, 256),
                                   (256, 384, 264), (512, 512, 1000), (300, 640, 520), (64, 1024, 8),
                                   (256, 272, 256), (300, 400, 264), (512, 1040, 256), (257, 3088, 520), (128, 368, 128)])   # round 3: K tails (K % 128 != 0, staged with per-lane masks)
@pytest.mark.parametrize("wk", [L.KERNEL_GEMM_256W, L.KERNEL_GEMM_256x128W])
def test_gemm256w_parity_and_bits_of_the_ring_kernel(native, cuda, oracle, M, K, N, wk):
    """The generated-assembly loop keeps the ring kernels' LDS image, fragment -> MFMA operand map and K order, so its
    result must equal the unsplit ring kernel's BIT FOR BIT (what makes a sharded linear equal the unsharded one does
    not depend on which tile kernel a shape lands on) and the oracle's within the matrix-core tolerance; every epilogue
    form, both loop variants (the NaN redo runs the scrubbing loop), K down to the two-step minimum, and any M: the rows
    / columns of ragged last tiles are read as zeros through the descriptors' range checks and their stores dropped."""
    rng = np.random.default_rng(M + K + N)
    A = clean_bytes(rng, (M, K))
    B = clean_bytes(rng, (N, K))
    sa = rng.uniform(0.005, 0.02, size=M).astype(np.float32)
    sb = rng.uniform(0.005, 0.02, size=N).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    tA, tB = dev(A, cuda), dev(B, cuda)

    def both(**kw):
        a = native.fp8_scaled_mm(tA, tB, kernel=wk, **kw)
        b = native.fp8_scaled_mm(tA, tB, kernel=L.KERNEL_GEMM_256, split_k=1, **kw)
        assert torch.equal(a.isnan(), b.isnan()) and torch.equal(a.nan_to_num(), b.nan_to_num()), kw.keys()
        return a

    check_mm(oracle, native, cuda, A, B, [0.01], [0.02], kernel=wk)
    check_mm(oracle, native, cuda, A, B, sa, sb, kernel=wk, bias=bias, scale_result=0.25)
    check_mm(oracle, native, cuda, A, B, sa, [0.02], kernel=wk, bias=bias, out_dtype=torch.bfloat16)
    check_mm(oracle, native, cuda, A, B, [0.01], sb, kernel=wk, out_dtype=torch.float16)
    s1, sN, sM = torch.full((1,), 0.01), dev(sb, cuda), dev(sa, cuda)
    for od in (torch.float32, torch.bfloat16, torch.float16):
        both(scale_a=s1, scale_b=s1, out_dtype=od)
        both(scale_a=sM, scale_b=sN, out_dtype=od, bias=dev(bias, cuda))
        both(scale_a=s1, scale_b=sN, out_dtype=od, bias=dev(bias, cuda, torch.bfloat16), scale_result=torch.full((1,), 0.5))
        both(scale_a=sM, scale_b=s1, out_dtype=od, bias=dev(bias, cuda, torch.float16))
    # transposed epilogue (bias per row of this call, scales in the swapped order)
    bias_m = rng.standard_normal(M).astype(np.float32)
    both(scale_a=sM, scale_b=sN, bias=dev(bias_m, cuda), transposed_epilogue=True)
    both(scale_a=s1, scale_b=s1, transposed_epilogue=True, out_dtype=torch.bfloat16)
    # NaN bytes: reference semantics (decode to 0: the tile is redone with the scrubbing loop) and OCP propagation
    A[min(3, M - 1), 17] = 0x7F
    B[N - 1, K - 1] = 0xFF
    tA, tB = dev(A, cuda), dev(B, cuda)
    check_mm(oracle, native, cuda, A, B, sa, sb, kernel=wk, bias=bias)
    both(scale_a=sM, scale_b=sN, bias=dev(bias, cuda), out_dtype=torch.bfloat16)
    got = both(scale_a=s1, scale_b=s1, nan_mode=L.NAN_PROPAGATE).cpu().numpy()
    nan = np.isnan(got)
    assert nan[min(3, M - 1), :].all() and nan[:, N - 1].all() and nan.sum() == N + M - 1


@pytest.mark.parametrize("M,K,N", [(512, 1024, 384), (300, 640, 264), (129, 3072, 72), (1024, 512, 256)])
def test_every_unsplit_tile_kernel_gives_the_same_bits(native, cuda, oracle, M, K, N):
    """What the sharded linear's bit equality rests on: whichever tile kernel a shape - or its transposed shard - lands on, an UNSPLIT tile kernel adds the
    K-steps of an output element in the same order, so all of them (ring kernels of every tile shape, the deep-ring 128x128, the small-batch tiles, both one-wave-per-SIMD
    forms) return identical bits for identical inputs and epilogues; one of them is checked against the oracle."""
    rng = np.random.default_rng(M + 3 * K + N)
    A = clean_bytes(rng, (M, K))
    B = clean_bytes(rng, (N, K))
    sa = rng.uniform(0.005, 0.02, size=M).astype(np.float32)
    sb = rng.uniform(0.005, 0.02, size=N).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    tA, tB, tsa, tsb, tb = dev(A, cuda), dev(B, cuda), dev(sa, cuda), dev(sb, cuda), dev(bias, cuda)
    check_mm(oracle, native, cuda, A, B, sa, sb, kernel=L.KERNEL_GEMM_128D, bias=bias)
    kernels = list(TILE_KERNELS) + [L.KERNEL_GEMM_256W, L.KERNEL_GEMM_256x128W]
    for od in (torch.float32, torch.bfloat16):
        ref = None
        for kern in kernels:
            got = native.fp8_scaled_mm(tA, tB, tsa, tsb, bias=tb, out_dtype=od, kernel=kern, split_k=1)
            if ref is None:
                ref = got
            assert torch.equal(got, ref), (kern, od)
        auto = native.fp8_scaled_mm(tA, tB, tsa, tsb, bias=tb, out_dtype=od, split_k=1)   # AUTO without a K split: one of them
        assert torch.equal(auto, ref), od


@pytest.mark.parametrize("wk", [L.KERNEL_GEMM_256W, L.KERNEL_GEMM_256x128W])
def test_gemm256w_random_shapes(native, cuda, oracle, wk):
    """Seeded random shapes inside the kernel's envelope (any M, N a multiple of 8, K a multiple of 128) with random
    epilogue forms, forced through the 256W kernel: oracle tolerance and the ring kernel's bits."""
    rng = np.random.default_rng(2026)
    for _ in range(10):
        M = int(rng.integers(1, 1100))
        N = 8 * int(rng.integers(1, 140))
        K = 128 * int(rng.integers(2, 12)) + (16 * int(rng.integers(1, 8)) if rng.integers(3) == 0 else 0)   # every third shape: a K tail
        A = clean_bytes(rng, (M, K))
        B = clean_bytes(rng, (N, K))
        sa = rng.uniform(0.005, 0.02, size=M).astype(np.float32) if rng.integers(2) else np.array([0.01], np.float32)
        sb = rng.uniform(0.005, 0.02, size=N).astype(np.float32) if rng.integers(2) else np.array([0.02], np.float32)
        bias = rng.standard_normal(N).astype(np.float32) if rng.integers(2) else None
        od = [torch.float32, torch.bfloat16, torch.float16][int(rng.integers(3))]
        if rng.integers(4) == 0:
            A[int(rng.integers(M)), int(rng.integers(K))] = 0x7F   # a NaN byte: the scrubbing redo
        got = check_mm(oracle, native, cuda, A, B, sa, sb, kernel=wk, bias=bias, out_dtype=od)
        kw = {"bias": dev(bias, cuda)} if bias is not None else {}
        ref = native.fp8_scaled_mm(dev(A, cuda), dev(B, cuda), dev(sa, cuda), dev(sb, cuda), out_dtype=od, kernel=L.KERNEL_GEMM_256,
                                   split_k=1, **kw)
        assert torch.equal(got, ref), (M, K, N, od)


@pytest.mark.parametrize("wk", [L.KERNEL_GEMM_256W, L.KERNEL_GEMM_256x128W])
@pytest.mark.parametrize("M,K,N", [(300, 384, 260), (256, 512, 12), (513, 640, 1020), (130, 400, 132), (256, 256, 4)])
def test_gemm256w_fp32_rows_with_n_a_multiple_of_4(native, cuda, oracle, M, K, N, wk):
    """fp32 output only needs N % 4 == 0 (16-byte stores).  The fp32 epilogue takes ONE 4-column chunk from each of two rows per lane (so that
    a store instruction writes whole rows): a last chunk that starts at N - 4 must be written, the next one dropped - ragged N that is not a
    multiple of 8, every epilogue form, against the oracle and the ring kernel's bits; the guard columns beyond N (ldc > N) stay untouched."""
    rng = np.random.default_rng(M * 7 + K + N)
    A = clean_bytes(rng, (M, K))
    B = clean_bytes(rng, (N, K))
    sa = rng.uniform(0.005, 0.02, size=M).astype(np.float32)
    sb = rng.uniform(0.005, 0.02, size=N).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    bias_m = rng.standard_normal(M).astype(np.float32)
    tA, tB = dev(A, cuda), dev(B, cuda)
    check_mm(oracle, native, cuda, A, B, [0.01], [0.02], kernel=wk)
    check_mm(oracle, native, cuda, A, B, sa, sb, kernel=wk, bias=bias, scale_result=0.25)
    s1, sN, sM = torch.full((1,), 0.01), dev(sb, cuda), dev(sa, cuda)
    forms = [dict(scale_a=s1, scale_b=s1), dict(scale_a=sM, scale_b=sN, bias=dev(bias, cuda)),
             dict(scale_a=s1, scale_b=sN, bias=dev(bias, cuda, torch.bfloat16), scale_result=torch.full((1,), 0.5)),
             dict(scale_a=sM, scale_b=sN, bias=dev(bias_m, cuda), transposed_epilogue=True)]
    for kw in forms:
        a = native.fp8_scaled_mm(tA, tB, kernel=wk, **kw)
        b = native.fp8_scaled_mm(tA, tB, kernel=L.KERNEL_GEMM_256, split_k=1, **kw)
        assert torch.equal(a, b), kw.keys()
        wide = torch.full((M, N + 12), -7.0, device=cuda)          # ldc = N + 12: the columns beyond N belong to the caller
        native.fp8_scaled_mm(tA, tB, kernel=wk, out=wide[:, :N], **kw)
        assert torch.equal(wide[:, :N], b) and bool((wide[:, N:] == -7.0).all()), kw.keys()


def test_gemm256w_envelope_dispatch_and_exactness(native, cuda, oracle):
    """Outside whole tiles / K-steps the explicit id refuses and AUTO falls back to the ring kernels; inside, AUTO picks it
    for large shapes (same bits either way); operands within a 2^12 product range are summed exactly by the matrix core,
    so a wrong register or LDS address in the generated loop cannot hide behind the hardware tolerance; a padded row
    stride (lda, ldb, ldc > the row length) goes through the descriptors and the epilogue untouched."""
    z = torch.zeros
    for (M, K, N) in ((256, 256, 301), (256, 128, 256), (256, 144, 256), (256, 200, 256)):   # (a K tail needs three K-steps: K > 256)
        with pytest.raises(RuntimeError):
            native.fp8_scaled_mm(z(M, K, dtype=torch.uint8, device=cuda), z(N, K, dtype=torch.uint8, device=cuda), torch.ones(1),
                                 torch.ones(1), kernel=L.KERNEL_GEMM_256W)
    with pytest.raises(RuntimeError):   # a forced K split belongs to the ring kernels
        native.fp8_scaled_mm(z(256, 4096, dtype=torch.uint8, device=cuda), z(256, 4096, dtype=torch.uint8, device=cuda), torch.ones(1),
                             torch.ones(1), kernel=L.KERNEL_GEMM_256W, split_k=2)
    rng = np.random.default_rng(79)
    A = (0x28 + rng.integers(0, 0x20, size=(512, 1024))).astype(np.uint8)
    B = (0x28 + rng.integers(0, 0x20, size=(768, 1024))).astype(np.uint8)
    check_mm(oracle, native, cuda, A, B, [0.5], [2.0], kernel=L.KERNEL_GEMM_256W, tol=MM_TOL)
    check_mm(oracle, native, cuda, A, B, [0.5], [2.0], kernel=L.KERNEL_GEMM_256x128W, tol=MM_TOL)
    # AUTO on a large shape of whole tiles: the same bits as the explicit id and as the ring kernel
    g = torch.Generator(device=cuda).manual_seed(5)
    a = torch.randint(0, 0x7F, (2048, 1024), dtype=torch.uint8, device=cuda, generator=g)
    b = torch.randint(0, 0x7F, (4096, 1024), dtype=torch.uint8, device=cuda, generator=g)
    s = torch.full((1,), 0.01)
    r_auto = native.fp8_scaled_mm(a, b, s, s, out_dtype=torch.bfloat16)
    r_w = native.fp8_scaled_mm(a, b, s, s, out_dtype=torch.bfloat16, kernel=L.KERNEL_GEMM_256W)
    r_ring = native.fp8_scaled_mm(a, b, s, s, out_dtype=torch.bfloat16, kernel=L.KERNEL_GEMM_256, split_k=1)
    r_n = native.fp8_scaled_mm(a, b, s, s, out_dtype=torch.bfloat16, kernel=L.KERNEL_GEMM_256x128W)   # (what AUTO picks for this shape: 256 tiles of 256x128)
    assert torch.equal(r_auto, r_w) and torch.equal(r_w, r_ring) and torch.equal(r_n, r_ring)
    # padded strides
    bufA = torch.randint(0, 0x7F, (256, 640), dtype=torch.uint8, device=cuda, generator=g)
    bufB = torch.randint(0, 0x7F, (512, 768), dtype=torch.uint8, device=cuda, generator=g)
    out = torch.full((256, 1024), -7.0, dtype=torch.float32, device=cuda)
    va, vb, vo = bufA[:, :512], bufB[:, :512], out[:, :512]
    native.fp8_scaled_mm(va, vb, s, s, kernel=L.KERNEL_GEMM_256W, out=vo)
    ref = native.fp8_scaled_mm(va.contiguous(), vb.contiguous(), s, s, kernel=L.KERNEL_GEMM_256, split_k=1)
    assert torch.equal(vo, ref) and bool((out[:, 512:] == -7.0).all())
    # ragged M: the rows of the last m-tile beyond M are not written (canary rows behind the output)
    big = torch.full((512, 256), -7.0, dtype=torch.bfloat16, device=cuda)
    a3 = torch.randint(0, 0x7F, (300, 384), dtype=torch.uint8, device=cuda, generator=g)
    b3 = torch.randint(0, 0x7F, (256, 384), dtype=torch.uint8, device=cuda, generator=g)
    native.fp8_scaled_mm(a3, b3, s, s, kernel=L.KERNEL_GEMM_256W, out=big[:300], out_dtype=torch.bfloat16)
    ref3 = native.fp8_scaled_mm(a3, b3, s, s, kernel=L.KERNEL_GEMM_256, split_k=1, out_dtype=torch.bfloat16)
    assert torch.equal(big[:300], ref3) and bool((big[300:] == -7.0).all())
    # ragged N inside a wider buffer: the columns beyond N keep the canary
    wide = torch.full((300, 512), -7.0, dtype=torch.bfloat16, device=cuda)
    b4 = torch.randint(0, 0x7F, (264, 384), dtype=torch.uint8, device=cuda, generator=g)
    native.fp8_scaled_mm(a3, b4, s, s, kernel=L.KERNEL_GEMM_256W, out=wide[:, :264], out_dtype=torch.bfloat16)
    ref4 = native.fp8_scaled_mm(a3, b4, s, s, kernel=L.KERNEL_GEMM_256, split_k=1, out_dtype=torch.bfloat16)
    assert torch.equal(wide[:, :264], ref4) and bool((wide[:, 264:] == -7.0).all())
    with pytest.raises(RuntimeError):   # 16-bit rows are stored 8 columns at a time
        native.fp8_scaled_mm(a3, torch.zeros(260, 384, dtype=torch.uint8, device=cuda), s, s, kernel=L.KERNEL_GEMM_256W, out_dtype=torch.bfloat16)


@pytest.mark.parametrize("kernel", TILE_KERNELS)
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_full_tile_staged_epilogue(native, cuda, oracle, kernel, out_dtype):
    """Interior (full) tiles take the LDS-staged, line-coalesced epilogue: per-row
    scales, bias and result scale must land on the right rows / columns in every
    tile variant and output type."""
    rng = np.random.default_rng(55)
    M, K, N = 512, 384, 768
    A = clean_bytes(rng, (M, K))
    B = clean_bytes(rng, (N, K))
    sa = rng.uniform(0.005, 0.02, size=M).astype(np.float32)
    sb = rng.uniform(0.005, 0.02, size=N).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    check_mm(oracle, native, cuda, A, B, sa, sb, kernel=kernel, bias=bias, scale_result=0.25, out_dtype=out_dtype)


# ---------------------------------------------------------------------------
# split-K (no counterpart in the reference: include/fp8mi.h, fp8mi_scaled_mm_ws)
# ---------------------------------------------------------------------------
def _counters_zero(native, cuda):
    ws = native._workspace(cuda)
    torch.cuda.synchronize()
    return int(ws[:L.WS_COUNTER_BYTES].view(torch.int32).abs().sum().item()) == 0


@pytest.mark.parametrize("M,K,N,kernel,split", [
    (128, 4096, 512, L.KERNEL_GEMM_128x64, 4), (64, 2048, 256, L.KERNEL_GEMM_64x128, 0),
    (100, 2992, 200, L.KERNEL_GEMM_128x64, 3),     # ragged tile, K tail in the last slice
    (40, 1040, 130, L.KERNEL_GEMM_64x128, 5),      # more slices asked for than K has ring stages of 256 B
    (130, 4096, 70, L.KERNEL_GEMM_128x64, 16), (256, 4096, 1024, L.KERNEL_AUTO, 0),
    (300, 2048, 300, L.KERNEL_GEMM_128, 2), (512, 1024, 512, L.KERNEL_GEMM_256, 2),
    (64, 14336, 512, L.KERNEL_GEMM_64x64, 0), (50, 3584, 200, L.KERNEL_GEMM_64x64, 4), (96, 4096, 320, L.KERNEL_GEMM_64x64, 7),   # round 3: the decode regime's tiles
    (32, 14336, 512, L.KERNEL_GEMM_32x64, 0), (9, 2992, 130, L.KERNEL_GEMM_32x64, 3), (40, 8192, 192, L.KERNEL_GEMM_32x64, 16),
    (32, 8192, 512, L.KERNEL_GEMM_32x32, 0), (20, 2992, 100, L.KERNEL_GEMM_32x32, 5), (70, 4096, 96, L.KERNEL_GEMM_32x32, 2), (16, 8192, 8192, L.KERNEL_AUTO, 0),
    (24, 12288, 3072, L.KERNEL_AUTO, 0), (64, 4096, 4096, L.KERNEL_AUTO, 0), (96, 4096, 4096, L.KERNEL_AUTO, 0),                   # ... as AUTO picks them
    (16, 8192, 4096, L.KERNEL_AUTO, 0),            # M <= 32 with a large weight matrix: auto leaves the skinny kernel
    (33, 14336, 512, L.KERNEL_AUTO, 0)])
def test_split_k_parity_and_reproducibility(native, cuda, oracle, M, K, N, kernel, split):
    """K cut into slices, fp32 partial tiles summed in slice order by the last workgroup of each tile:
    parity with the oracle, bit-identical from run to run, and the tile counters are left zero."""
    rng = np.random.default_rng(M + K + N + split)
    A, B = clean_bytes(rng, (M, K)), clean_bytes(rng, (N, K))
    sa = rng.uniform(0.005, 0.02, size=M).astype(np.float32)
    sb = rng.uniform(0.005, 0.02, size=N).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    for od in (torch.float32, torch.bfloat16):
        got = check_mm(oracle, native, cuda, A, B, sa, sb, kernel=kernel, bias=bias, out_dtype=od, split_k=split,
                       tol=MFMA_TOL)
        again = native.fp8_scaled_mm(dev(A, cuda), dev(B, cuda), dev(sa, cuda), dev(sb, cuda), bias=dev(bias, cuda),
                                     out_dtype=od, kernel=kernel, split_k=split)
        assert torch.equal(got, again)
    assert _counters_zero(native, cuda)


def test_split_k_nan_modes(native, cuda, oracle):
    """A NaN byte inside one K slice: reference mode redoes only that slice's partial with the scrub;
    OCP mode poisons the row / column through the partial sum."""
    rng = np.random.default_rng(77)
    A, B = clean_bytes(rng, (96, 2048)), clean_bytes(rng, (160, 2048))
    A[3, 1500] = 0x7F
    B[5, 10] = 0xFF
    check_mm(oracle, native, cuda, A, B, [1.0], [1.0], kernel=L.KERNEL_GEMM_128x64, split_k=4)
    got = native.fp8_scaled_mm(dev(A, cuda), dev(B, cuda), torch.ones(1), torch.ones(1), kernel=L.KERNEL_GEMM_128x64,
                               nan_mode=L.NAN_PROPAGATE, split_k=4).cpu().numpy()
    nan = np.isnan(got)
    assert nan[3, :].all() and nan[:, 5].all() and nan.sum() == 160 + 96 - 1
    assert _counters_zero(native, cuda)


def test_split_k_workspace_contract(native, cuda, oracle):
    """C ABI: a missing, misaligned or too-small workspace silently disables the split (same bits as split_k = 1);
    split_k < 0 is an argument error; the advertised size covers the library's own choices."""
    lib = L.load()
    assert lib.fp8mi_scaled_mm_workspace_bytes() >= L.WS_COUNTER_BYTES + 256 * 128 * 64 * 4
    rng = np.random.default_rng(78)
    M, K, N = 128, 4096, 256
    A, B = dev(clean_bytes(rng, (M, K)), cuda), dev(clean_bytes(rng, (N, K)), cuda)
    s1 = torch.full((1,), 0.01, device=cuda)
    st = torch.cuda.current_stream().cuda_stream

    def run(ws_ptr, ws_bytes, split):
        C = torch.empty(M, N, device=cuda)
        rc = lib.fp8mi_scaled_mm_ws(A.data_ptr(), B.data_ptr(), C.data_ptr(), s1.data_ptr(), s1.data_ptr(), None, None,
                                    M, N, K, K, K, N, 0, 0, L.F32, 0, 0, L.KERNEL_GEMM_128x64, split, ws_ptr, ws_bytes, st)
        torch.cuda.synchronize()
        return rc, C

    ws = torch.zeros(int(lib.fp8mi_scaled_mm_workspace_bytes()), dtype=torch.uint8, device=cuda)
    rc, base = run(None, 0, 1)
    assert rc == 0
    rc, split = run(ws.data_ptr(), ws.numel(), 4)
    assert rc == 0 and not torch.equal(split, base)            # really took the split path (different summation order)
    assert torch.allclose(split, base, rtol=0, atol=2e-3 * float(base.abs().max()))
    for ptr, nbytes in ((None, 0), (ws.data_ptr() + 4, ws.numel() - 4), (ws.data_ptr(), L.WS_COUNTER_BYTES + 1024),
                        (ws.data_ptr(), 16)):
        rc, c = run(ptr, nbytes, 4)
        assert rc == 0 and torch.equal(c, base)
    rc, _ = run(ws.data_ptr(), ws.numel(), -1)
    assert rc == -3 and b"split_k" in lib.fp8mi_last_error()
    assert int(ws[:L.WS_COUNTER_BYTES].view(torch.int32).abs().sum().item()) == 0


def test_split_k_full_size_decode_shape(native, cuda, oracle):
    """M = 64 rows against a 14336 x 4096 weight matrix (the C2 weight shape with a small batch):
    auto dispatch slices K; checked against the C oracle in float64."""
    rng = np.random.default_rng(79)
    A, B = clean_bytes(rng, (64, 14336)), clean_bytes(rng, (4096, 14336))
    got = check_mm(oracle, native, cuda, A, B, [0.01], [0.02], out_dtype=torch.float32)
    one = native.fp8_scaled_mm(dev(A, cuda), dev(B, cuda), torch.full((1,), 0.01), torch.full((1,), 0.02), split_k=1)
    assert not torch.equal(got, one)  # the auto path did split
    assert _counters_zero(native, cuda)


def test_random_shapes_auto_dispatch_fuzz(native, cuda, oracle):
    """Seeded sweep of 80 random problems through the automatic dispatch (GEMV / skinny / tile kernels with and
    without split-K / generic), random scale layouts, bias, output type and split_k request: every output element
    against the oracle.  Catches tile-map, K-slice and edge-masking corner cases the hand-picked shapes miss."""
    rng = np.random.default_rng(2024)
    pick = lambda xs: xs[int(rng.integers(len(xs)))]
    for it in range(80):
        M = int(pick([1, 2, 3, 7, 16, 31, 33, 48, 64, 65, 100, 128, 129, 200, 256, 300, 511, 640]))
        N = int(pick([1, 5, 16, 63, 64, 65, 127, 128, 130, 255, 256, 384, 500, 777, 1024]))
        K = int(pick([16, 32, 112, 128, 144, 256, 272, 512, 1040, 2048, 4096, 4112, 6144])) if rng.random() < 0.9 else int(rng.integers(1, 300))
        if M * N * K > 6.0e8:   # keep the float64 oracle quick
            K = 512
        A, B = clean_bytes(rng, (M, K)), clean_bytes(rng, (N, K))
        sa = rng.uniform(0.005, 0.02, size=M if rng.random() < 0.5 else 1).astype(np.float32)
        sb = rng.uniform(0.005, 0.02, size=N if rng.random() < 0.5 else 1).astype(np.float32)
        bias = rng.normal(size=N).astype(np.float32) if rng.random() < 0.5 else None
        od = pick([torch.float32, torch.bfloat16, torch.float16])
        split = int(pick([0, 0, 0, 1, 2, 3, 5, 8]))
        try:
            check_mm(oracle, native, cuda, A, B, sa, sb, bias=bias, out_dtype=od, split_k=split, tol=MFMA_TOL)
        except AssertionError as e:
            raise AssertionError(f"case {it}: M={M} K={K} N={N} sa={sa.size} sb={sb.size} bias={bias is not None} "
                                 f"out={od} split_k={split}: {e}") from e
    assert _counters_zero(native, cuda)


def test_graph_capture_of_the_c_abi(native, cuda, oracle):
    """Entry points only enqueue (no sync, no allocation): capturable in a HIP graph."""
    rng = np.random.default_rng(56)
    A, B = clean_bytes(rng, (64, 256)), clean_bytes(rng, (128, 256))
    a, b = dev(A, cuda), dev(B, cuda)
    s1 = torch.full((1,), 0.5, device=cuda)
    x = torch.randn(4096, device=cuda)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        native.fp8_scaled_mm(a, b, s1, s1)
        native.fp8_quantize(x)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    A2, B2 = clean_bytes(rng, (64, 4096)), clean_bytes(rng, (256, 4096))   # split-K path (workspace + counters)
    a2, b2 = dev(A2, cuda), dev(B2, cuda)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        c = native.fp8_scaled_mm(a, b, s1, s1)
        c2 = native.fp8_scaled_mm(a2, b2, s1, s1)
        q, inv = native.fp8_quantize(x)
        h = native.fp8_dequantize(q, inv)
    c.zero_(); q.zero_(); c2.zero_()
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    exact2 = oracle.scaled_mm(A2, B2, [0.5], [0.5], accumulate="f64")
    assert np.all(np.abs(c2.cpu().numpy() - exact2) <= MFMA_TOL * oracle.abs_dot_bound(A2, B2, [0.5], [0.5]))
    exact = oracle.scaled_mm(A, B, [0.5], [0.5], accumulate="f64")
    assert np.all(np.abs(c.cpu().numpy() - exact) <= MFMA_TOL * oracle.abs_dot_bound(A, B, [0.5], [0.5]))
    eq, einv = oracle.quantize(x.cpu().numpy())
    assert np.array_equal(q.cpu().numpy(), eq) and float(inv.cpu()) == float(einv)
    assert h.shape == x.shape


def test_empty_and_degenerate(native, cuda):
    z = lambda *s: torch.zeros(*s, dtype=torch.uint8, device=cuda)
    one = torch.ones(1)
    assert native.fp8_scaled_mm(z(0, 64), z(8, 64), one, one).shape == (0, 8)
    assert native.fp8_scaled_mm(z(4, 64), z(0, 64), one, one).shape == (4, 0)
    out = native.fp8_scaled_mm(z(4, 0), z(8, 0), one, one)
    assert out.shape == (4, 8) and not out.any()
    with pytest.raises(AssertionError):
        native.fp8_scaled_mm(z(4, 64), z(8, 32), one, one)
    with pytest.raises(AssertionError):
        native.fp8_scaled_mm(z(4, 64).float(), z(8, 64), one, one)
    with pytest.raises(AssertionError):
        native.fp8_scaled_mm(z(4, 64), z(8, 64), torch.ones(3), one)
    with pytest.raises(L.Fp8miError):
        native.fp8_scaled_mm(z(4, 64), z(8, 64), one, one, kernel=L.KERNEL_GEMV)


# ---------------------------------------------------------------------------
# the reference's own accuracy tests, restated (test_fp8_metal.py:97-218)
# ---------------------------------------------------------------------------

@pytest.mark.parametrize("M,K,N", [(64, 256, 128), (128, 256, 64), (1, 512, 256), (1, 4096, 4096), (4, 4096, 4096)])
def test_matmul_accuracy_vs_fp32(native, cuda, oracle, M, K, N):
    g = torch.Generator().manual_seed(1234)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g)
    ref = (A @ B.T).numpy()
    Aq, sa = native.fp8_quantize(A.to(cuda))
    Bq, sb = native.fp8_quantize(B.to(cuda))
    out = native.fp8_scaled_mm(Aq, Bq, sa, sb)
    assert oracle.rel_rmse(out.cpu().numpy(), ref) < REL_RMSE_GATE
    # and the GPU agrees with the CPU oracle run on the same bytes, to fp32 rounding
    exact = oracle.scaled_mm(Aq.cpu().numpy(), Bq.cpu().numpy(), sa.cpu().numpy(), sb.cpu().numpy(), accumulate="f64")
    bound = oracle.abs_dot_bound(Aq.cpu().numpy(), Bq.cpu().numpy(), sa.cpu().numpy(), sb.cpu().numpy())
    assert np.all(np.abs(out.cpu().numpy() - exact) <= (MFMA_TOL if M > 1 else MM_TOL) * bound + 1e-30)
    if M > 1:  # amax-quantised gaussians: the MFMA result is far closer than its worst case
        assert np.sqrt(np.mean((out.cpu().numpy() - exact) ** 2)) <= 2e-5 * np.sqrt(np.mean(bound ** 2))
    assert torch.equal(native.fp8_scaled_mm_auto(Aq, Bq, sa, sb), out)
    assert torch.equal(native.fp8_scaled_mm_fast(Aq, Bq, sa, sb), out)


# ---------------------------------------------------------------------------
# size-independent properties at BASELINE sizes
# ---------------------------------------------------------------------------

def test_properties_flux_shape(native, cuda):
    """FLUX linear M=4096, K=3072, N=12288 (BASELINE configs[3]): power-of-two
    scale linearity is exact, column shards reproduce the full product bit for
    bit (what the N-sharded multi-GPU path relies on), K-splitting is additive."""
    g = torch.Generator(device=cuda).manual_seed(7)
    M, K, N = 4096, 3072, 12288
    A = torch.randint(0, 256, (M, K), dtype=torch.uint8, device=cuda, generator=g)
    B = torch.randint(0, 256, (N, K), dtype=torch.uint8, device=cuda, generator=g)
    A[(A & 0x7F) == 0x7F] = 0x3C
    B[(B & 0x7F) == 0x7F] = 0x3C
    one = torch.ones(1, device=cuda)
    full = native.fp8_scaled_mm(A, B, one, one)
    assert torch.isfinite(full).all()
    assert torch.equal(native.fp8_scaled_mm(A, B, one * 4, one * 0.5), full * 2)
    shard = native.fp8_scaled_mm(A, B[1536:3072], one, one, kernel=L.KERNEL_GEMM_256)
    assert torch.equal(shard, native.fp8_scaled_mm(A, B, one, one, kernel=L.KERNEL_GEMM_256)[:, 1536:3072])
    h = 1536
    parts = native.fp8_scaled_mm(A[:, :h], B[:, :h], one, one) + native.fp8_scaled_mm(A[:, h:], B[:, h:], one, one)
    mag = native.fp8_scaled_mm(A & 0x7F, B & 0x7F, one, one)  # sum |a||b|
    assert bool(((parts - full).abs() <= 2 * MFMA_TOL * mag + 1e-30).all())
    # the transposed problem gives the transposed result
    tr = native.fp8_scaled_mm(B[:2048], A[:1024], one, one)
    assert bool(((tr.t() - full[:1024, :2048]).abs() <= 2 * MFMA_TOL * mag[:1024, :2048]).all())


def test_gemv_equals_gemm_row(native, cuda):
    """M == 1 through the GEMV kernel vs the same row through the MFMA kernel."""
    g = torch.Generator(device=cuda).manual_seed(8)
    K, N = 14336, 4096
    x = torch.randint(0, 126, (1, K), dtype=torch.uint8, device=cuda, generator=g)
    W = torch.randint(0, 126, (N, K), dtype=torch.uint8, device=cuda, generator=g)
    s = torch.tensor([0.01], device=cuda)
    a = native.fp8_scaled_mm(x, W, s, s, kernel=L.KERNEL_GEMV)
    b = native.fp8_scaled_mm(x, W, s, s, kernel=L.KERNEL_GEMM_128)
    mag = a.abs()  # all operands are non-negative here: sum|a||b| = the result
    assert bool(((a - b).abs() <= MFMA_TOL * mag).all())
