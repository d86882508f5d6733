"""GPU (MI355X): the two call-site rows of SURVEY.md 8f-2 and BASELINE config C5 at its stated size.

  * the ComfyUI plugin entry (reference __init__.py:13-61) imported as a module
    installs the patch; a ComfyUI-style fp8 `nn.Linear` whose weight is loaded by
    `.copy_()` from CPU fp8 bytes and by `.to(float8_e4m3fn)` (the weight-load
    scenarios of fp8_mps_patch.py:160-223, 250-291) runs the FLUX projections
    3072->12288, 12288->3072 and 3072->9216 at M = 4096, and every result is
    checked against the ORACLE on the same bytes (a row / column sample of the
    float64 product), not only against a float32 matmul;
  * C5: 2^30 elements float32 -> e4m3fn -> float16 (fp8_matmul.metal:215-236),
    sampled against the oracle across the whole range (last elements and byte
    offsets beyond 2^32 included - the reference's `uint count` stops at 2^32-1,
    metal:218,231) plus the idempotence property over all of it.
"""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

from conftest import PKG

pytestmark = pytest.mark.gpu

F8 = torch.float8_e4m3fn
MFMA_TOL = 1.0e-3  # see tests/test_gpu_parity.py: fp8 MFMA accumulation bound relative to sum|a||b|


def _import_plugin():
    """Import fp8-mps-metal_amd/__init__.py the way ComfyUI's custom-node loader does
    (importlib on the directory's __init__.py; the directory name has dashes)."""
    name = "fp8_mi355x_plugin_under_test"
    spec = importlib.util.spec_from_file_location(name, os.path.join(PKG, "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class ComfyStyleFP8Linear(torch.nn.Module):
    """What ComfyUI's fp8 ops do around torch._scaled_mm (the L5 call site of
    SURVEY.md 3.2): weight (N,K) float8_e4m3fn, activations clamped to +-448 and
    cast with .to(float8_e4m3fn), `_scaled_mm(x8, w.t(), bias, scales, out_dtype)`."""

    def __init__(self, in_features, out_features, device, bias=True):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.empty(out_features, in_features, dtype=F8, device=device),
                                         requires_grad=False)
        self.bias = torch.nn.Parameter(torch.zeros(out_features, dtype=torch.bfloat16, device=device),
                                       requires_grad=False) if bias else None
        self.scale_weight = torch.ones((), dtype=torch.float32, device=device)
        self.scale_input = torch.ones((), dtype=torch.float32, device=device)

    def forward(self, x):
        lead = x.shape[:-1]
        x2 = torch.clamp(x, min=-448, max=448).reshape(-1, x.shape[-1]).to(F8)
        y = torch._scaled_mm(x2, self.weight.t(), out_dtype=x.dtype, bias=self.bias,
                             scale_a=self.scale_input, scale_b=self.scale_weight)
        return y.reshape(*lead, self.weight.shape[0])


def _check_against_oracle(oracle, y, x8_bytes, w8_bytes, bias, rng, sa=1.0, sb=1.0):
    """|y - exact| <= MFMA_TOL * sum|a||b| + bf16 rounding on a sample of rows x columns
    (float64 oracle product of the SAME bytes), corners included."""
    M, N = y.shape
    rows = np.unique(np.concatenate([[0, 1, M - 1], rng.integers(0, M, 45)]))
    cols = np.unique(np.concatenate([[0, 1, N - 1], rng.integers(0, N, 253)]))
    A = x8_bytes[torch.from_numpy(rows).to(x8_bytes.device)].cpu().numpy()
    B = w8_bytes[torch.from_numpy(cols).to(w8_bytes.device)].cpu().numpy()
    exact = oracle.scaled_mm(A, B, [sa], [sb], accumulate="f64")
    bound = oracle.abs_dot_bound(A, B, [sa], [sb])
    if bias is not None:
        b = bias.float().cpu().numpy().astype(np.float64)[cols]
        exact, bound = exact + b[None, :], bound + np.abs(b)[None, :]
    got = y[torch.from_numpy(rows).to(y.device)][:, torch.from_numpy(cols).to(y.device)].float().cpu().numpy().astype(np.float64)
    eps = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11, torch.float32: 0.0}[y.dtype]
    err = np.abs(got - exact)
    assert np.all(err <= MFMA_TOL * bound + eps * np.abs(exact) + 1e-30), \
        f"max err/bound {np.max(err / (bound + 1e-300)):.3e}"
    return exact, got


def test_plugin_entry_installs_patch_and_flux_linears_match_oracle(cuda, oracle):
    import fp8_mps_patch
    fp8_mps_patch.uninstall()
    assert not fp8_mps_patch.is_installed()
    plugin = _import_plugin()                       # reference __init__.py:22-27: install() on import
    try:
        assert fp8_mps_patch.is_installed()
        assert torch._scaled_mm is fp8_mps_patch._metal_scaled_mm
        assert torch.Tensor.to is fp8_mps_patch._metal_tensor_to
        assert torch.Tensor.copy_ is fp8_mps_patch._metal_tensor_copy
        assert plugin.NODE_CLASS_MAPPINGS == {} and plugin.NODE_DISPLAY_NAME_MAPPINGS == {}   # :57-61
        assert set(plugin.__all__) == {"NODE_CLASS_MAPPINGS", "NODE_DISPLAY_NAME_MAPPINGS"}

        M = 4096
        g = torch.Generator().manual_seed(11)
        rng = np.random.default_rng(11)
        x_by_k = {}
        for (K, N, load) in ((3072, 12288, "copy_cpu_fp8"), (12288, 3072, "to_fp8"), (3072, 9216, "copy_bf16")):
            lin = ComfyStyleFP8Linear(K, N, cuda)
            w = torch.randn(N, K, generator=g) * 0.04
            if load == "copy_cpu_fp8":
                # checkpoint holds fp8 bytes on the CPU: _metal_tensor_copy "bytes" scenario (fp8_mps_patch.py:250-267)
                w8_cpu = torch.from_numpy(oracle.encode(w.numpy())).view(F8)
                lin.weight.data.copy_(w8_cpu)
                assert torch.equal(lin.weight.data.view(torch.uint8).cpu(), w8_cpu.view(torch.uint8))
            elif load == "to_fp8":
                # bf16 weights cast on the device: _metal_tensor_to "encode" scenario (:178-196)
                wb = w.to(torch.bfloat16)
                lin.weight.data = wb.to(cuda).to(F8)
                assert np.array_equal(lin.weight.data.view(torch.uint8).cpu().numpy(), oracle.encode(wb.float().numpy()))
            else:
                # bf16 CPU source copied into the fp8 parameter: "encode" scenario of copy_ (:271-291)
                wb = w.to(torch.bfloat16)
                lin.weight.data.copy_(wb)
                assert np.array_equal(lin.weight.data.view(torch.uint8).cpu().numpy(), oracle.encode(wb.float().numpy()))
            bias = (torch.randn(N, generator=g) * 0.1).to(torch.bfloat16)
            lin.bias.data.copy_(bias)
            if K not in x_by_k:
                x_by_k[K] = (torch.randn(M, K, generator=g) * 1.5).to(torch.bfloat16).to(cuda)
            x = x_by_k[K]
            y = lin(x)
            torch.cuda.synchronize()
            assert y.shape == (M, N) and y.dtype == torch.bfloat16 and y.device.type == "cuda"
            x8 = torch.clamp(x, min=-448, max=448).to(F8).view(torch.uint8)
            w8 = lin.weight.data.view(torch.uint8)
            exact, got = _check_against_oracle(oracle, y, x8, w8, lin.bias.data, rng)
            # ... and the reference's own accuracy gate against the unquantised fp32 linear (README.md:86): ~4 %
            ref = (x.float() @ w.to(cuda).t() + bias.to(cuda).float()[None, :])
            rr = oracle.rel_rmse(y.float().cpu().numpy(), ref.cpu().numpy())
            assert rr < 0.06, rr
            # a 3-D activation batch (B, S, K) through the same module
            y3 = lin(x[:128].reshape(2, 64, K))
            assert y3.shape == (2, 64, N) and torch.equal(y3.reshape(128, N), lin(x[:128]))
            del lin, y, ref
    finally:
        fp8_mps_patch.uninstall()
    assert not fp8_mps_patch.is_installed()


def test_padded_weight_rows_give_the_same_bits_through_both_entry_points(native, cuda, oracle):
    """native.pad_weight_rows: a weight copied once into rows K + 256 bytes apart (few-token calls against some power-of-two row
    strides crowd onto a few memory channels: profiles/r03_row_stride.txt).  The stride goes down to the C ABI as `ldb`: same bits as the
    unpadded call through native.fp8_scaled_mm and through the patched torch._scaled_mm (column-major `other` = padded.t()),
    for the few-rows, small-tile and large-tile regimes; oracle-checked once."""
    import fp8_mps_patch
    rng = np.random.default_rng(808)
    K, N = 8192, 1024
    W = torch.from_numpy(rng.integers(0, 120, size=(N, K), dtype=np.uint8)).to(cuda)
    Wp = native.pad_weight_rows(W)
    assert Wp.shape == W.shape and Wp.stride() == (K + 256, 1) and torch.equal(Wp, W)
    s = torch.full((1,), 0.01, device=cuda)
    fp8_mps_patch.install()
    try:
        for M in (1, 4, 16, 64, 300):
            X = torch.from_numpy(rng.integers(0, 120, size=(M, K), dtype=np.uint8)).to(cuda)
            a = native.fp8_scaled_mm(X, W, s, s, out_dtype=torch.bfloat16)
            b = native.fp8_scaled_mm(X, Wp, s, s, out_dtype=torch.bfloat16)
            assert torch.equal(a, b), M
            c = torch._scaled_mm(X.view(F8), Wp.view(F8).t(), scale_a=s, scale_b=s, out_dtype=torch.bfloat16)
            assert torch.equal(a, c), M
        _check_against_oracle(oracle, b, X, W, None, rng, sa=0.01, sb=0.01)
    finally:
        fp8_mps_patch.uninstall()


def test_c5_quantize_dequant_at_2_pow_30(native, cuda, oracle):
    """BASELINE config C5 at its stated size.  x = randn * 16, seed 1234."""
    n = 1 << 30
    extra = 4099                                    # crosses byte offset 2^32 of the fp32 input
    g = torch.Generator(device=cuda).manual_seed(1234)
    x = torch.empty(n + extra, dtype=torch.float32, device=cuda)
    chunk = 1 << 27
    for i in range(0, n + extra, chunk):
        m = min(chunk, n + extra - i)
        x[i:i + m] = torch.randn(m, device=cuda, generator=g) * 16
    q = native.fp8_encode(x[:n])                    # 2^30 elements: 4 GiB in, 1 GiB out
    h = native.fp8_dequantize(q, None)              # 1 GiB in, 2 GiB out
    torch.cuda.synchronize()
    assert q.numel() == n and h.numel() == n and h.dtype == torch.float16

    # 2^22 sampled indices over the whole range + both ends
    idx = torch.cat([torch.randint(0, n, (1 << 22,), device=cuda, generator=g),
                     torch.arange(0, 64, device=cuda), torch.arange(n - 64, n, device=cuda),
                     torch.arange((1 << 29) - 32, (1 << 29) + 32, device=cuda)])
    xs = x[idx].cpu().numpy()
    qs = q[idx].cpu().numpy()
    assert np.array_equal(qs, oracle.encode(xs))
    assert np.array_equal(h[idx].cpu().view(torch.int16).numpy().view(np.uint16),
                          oracle.dequantize_f16(qs).view(np.uint16))

    # idempotence over ALL 2^30: enc(dec(enc(x))) == enc(x), modulo 0x80 -> -0.0 -> 0x00
    # (the reference's own round-trip exception, test_fp8_correctness.py:118-131)
    q2 = native.fp8_encode(h)
    assert torch.equal(torch.where(q == 0x80, torch.zeros_like(q), q), q2)
    assert not bool(((q & 0x7F) == 0x7F).any())    # the reference encoder never emits a NaN pattern
    del q2, h

    # beyond 2^32 bytes: the same kernels over n + 4099 elements (input bytes 2^32 + 16396) and a float32
    # dequant whose OUTPUT crosses 2^32 bytes; the tail past 2^30 elements is compared in full
    qx = native.fp8_encode(x)
    assert torch.equal(qx[:n], q)
    assert np.array_equal(qx[n:].cpu().numpy(), oracle.encode(x[n:].cpu().numpy()))
    del x, q
    f = native.fp8_dequantize(qx, None, out_dtype=torch.float32)
    tail = qx[n - 64:].cpu().numpy()
    assert np.array_equal(f[n - 64:].cpu().numpy().view(np.uint32), oracle.decode(tail).view(np.uint32))
    idx2 = torch.randint(0, n + extra, (1 << 20,), device=cuda, generator=g)
    assert np.array_equal(f[idx2].cpu().numpy().view(np.uint32), oracle.decode(qx[idx2].cpu().numpy()).view(np.uint32))
