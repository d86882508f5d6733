"""GPU (MI355X): the monkey-patch surface end to end, restating the reference's
integration tests (test_fp8_metal.py:318-705, test_mps_vs_cpu.py:283-357,
validate_fix.py:50-160) with "cuda" (PyTorch-ROCm) in the role of "mps"."""
import datetime
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

F8 = torch.float8_e4m3fn
REL_TOL = 0.15  # the reference's FP8_E4M3FN_RELATIVE_TOLERANCE (test_fp8_metal.py:32)


def bytes_of(t):
    return t.view(torch.uint8).cpu().numpy()


def test_to_fp8_shapes_dtypes_devices(patch, native, cuda, oracle):
    """test_fp8_metal.py:352-465."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 8, generator=g)
    xd = x.to(cuda)
    for y in (xd.to(F8), xd.to(dtype=F8), xd.to(cuda, F8), x.to(cuda, F8), x.to(device="cuda", dtype=F8),
              xd.to(torch.device("cuda", 0), dtype=F8, non_blocking=True)):
        assert y.dtype == F8 and y.device.type == "cuda" and y.shape == (4, 8)
        assert np.array_equal(bytes_of(y), oracle.encode(x.numpy()))
    for src in (torch.float16, torch.bfloat16, torch.float32):
        xs = torch.randn(4, 4, generator=g).to(src)
        y = xs.to(cuda).to(F8)
        assert np.array_equal(bytes_of(y), oracle.encode(xs.float().numpy()))
    e = torch.empty(0, device=cuda).to(F8)
    assert e.dtype == F8 and e.numel() == 0
    s = torch.tensor([3.14], device=cuda).to(F8)
    assert s.shape == (1,) and abs(float(s.to(torch.float32).cpu()) - 3.14) / 3.14 < REL_TOL
    big = torch.randn(128, 256, generator=g)
    assert np.array_equal(bytes_of(big.to(cuda).to(F8)), oracle.encode(big.numpy()))
    nc = torch.randn(16, 32, generator=g).to(cuda).t()  # non-contiguous source
    assert np.array_equal(bytes_of(nc.to(F8)), oracle.encode(nc.cpu().numpy()))


def test_value_preservation_no_auto_scaling(patch, native, cuda, golden_dir):
    """test_fp8_metal.py:582-705, validate_fix.py: .to() must not rescale; bytes
    equal the reference spec's for every value list the reference tests use."""
    kat = json.load(open(os.path.join(golden_dir, "kat.json")))
    for name, pairs in kat["value_lists"].items():
        x = torch.tensor([p[0] for p in pairs], dtype=torch.float32, device=cuda)
        y = x.to(F8)
        assert bytes_of(y).tolist() == [p[1] for p in pairs], name
        back = y.to(torch.float32)
        for v, b in zip(x.cpu().tolist(), back.cpu().tolist()):
            if abs(v) >= 0.015625:
                assert abs(b - min(max(v, -448.0), 448.0)) <= REL_TOL * abs(v), (name, v, b)
    one = torch.tensor([1.0], device=cuda).to(F8)
    assert float(one.to(torch.float32).cpu()) == 1.0  # 1.0 -> 1.0, not 448 (test_fp8_metal.py:660-675)


def test_torch_cpu_bytes_agree(patch, native, cuda):
    """test_mps_vs_cpu.py:283-357: device bytes == torch-CPU bytes for [0.5,1,2,10,100]."""
    v = torch.tensor([0.5, 1.0, 2.0, 10.0, 100.0])
    cpu_bytes = patch._original_tensor_to(v, F8).view(torch.uint8)
    assert torch.equal(v.to(cuda).to(F8).view(torch.uint8).cpu(), cpu_bytes)


def test_fp8_cpu_to_device_is_byte_exact(patch, native, cuda):
    """test_fp8_metal.py:467-483: pre-quantised CPU weights move as raw bytes."""
    raw = torch.arange(256, dtype=torch.uint8).repeat(3)
    w = raw.view(F8)
    for moved in (w.to(cuda), w.to("cuda"), w.to(device=cuda), w.to(cuda, F8), w.to(cuda, non_blocking=True)):
        assert moved.dtype == F8 and moved.device.type == "cuda"
        assert torch.equal(moved.view(torch.uint8).cpu(), raw)
    e5 = raw.view(torch.float8_e5m2).to(cuda)
    assert e5.dtype == torch.float8_e5m2 and torch.equal(e5.view(torch.uint8).cpu(), raw)
    # dtype given with the move: honoured (the reference drops it)
    h = w.to(cuda, torch.float16)
    assert h.dtype == torch.float16 and h.device.type == "cuda"


def test_fp8_on_device_to_float(patch, native, cuda, golden_dir):
    g = json.load(open(os.path.join(golden_dir, "decode_256.json")))
    w = torch.arange(256, dtype=torch.uint8, device=cuda).view(F8)
    assert w.to(F8) is w and w.to(cuda) is not None
    h = w.to(torch.float16)
    assert np.array_equal(h.cpu().view(torch.int16).numpy().view(np.uint16), np.array(g["f16_bits"], np.uint16))
    f = w.to(torch.float32)
    assert np.array_equal(f.cpu().numpy().view(np.uint32), np.array(g["f32_bits"], np.uint32))
    assert torch.equal(w.to(torch.bfloat16).float(), f)
    assert torch.equal(w.to(torch.float64).cpu(), f.double().cpu())
    assert torch.equal(w.to(dtype=torch.float32, device=cuda), f)
    assert w.reshape(16, 16).to(torch.float16).shape == (16, 16)
    back = w.to("cpu")  # device -> CPU is torch's own path
    assert back.device.type == "cpu" and torch.equal(back.view(torch.uint8), torch.arange(256, dtype=torch.uint8))


def test_copy_scenarios(patch, native, cuda, oracle):
    """test_fp8_metal.py:486-579."""
    raw = torch.arange(256, dtype=torch.uint8)
    dst = torch.empty(256, dtype=F8, device=cuda)
    assert dst.copy_(raw.view(F8)) is dst                      # CPU fp8 -> device fp8 (weight load)
    assert torch.equal(dst.view(torch.uint8).cpu(), raw)
    dst2 = torch.empty(256, dtype=F8, device=cuda)
    dst2.copy_(dst.flip(0))                                      # device fp8 -> device fp8 (stochastic rounding path)
    assert torch.equal(dst2.view(torch.uint8).cpu(), raw.flip(0))
    src = torch.tensor([[1.0, 2.5, -3.0, 0.5], [10.0, -8.0, 0.0, 100.0]], device=cuda)  # test_fp8_metal.py:527-528
    d3 = torch.empty(2, 4, dtype=F8, device=cuda)
    assert d3.copy_(src) is d3
    assert np.array_equal(bytes_of(d3), oracle.encode(src.cpu().numpy()))
    d4 = torch.empty(2, 4, dtype=F8, device=cuda)
    d4.copy_(src.cpu().to(torch.bfloat16))                       # CPU bf16 source
    assert np.array_equal(bytes_of(d4), oracle.encode(src.cpu().numpy()))
    d5 = torch.empty(3, 4, dtype=F8, device=cuda)
    d5.copy_(torch.tensor([1.0, 5.0, 10.0, 50.0], device=cuda))  # broadcast (validate_fix.py:110)
    assert bytes_of(d5).tolist() == [oracle.encode(np.array([1.0, 5.0, 10.0, 50.0], np.float32)).tolist()] * 3
    f = torch.zeros(4, device=cuda)
    f.copy_(torch.ones(4))                                       # nothing fp8: untouched path
    assert float(f.sum().cpu()) == 4.0


def test_patched_scaled_mm(patch, native, cuda, oracle):
    g = torch.Generator().manual_seed(5)
    A = torch.randn(16, 32, generator=g)
    B = torch.randn(32, 32, generator=g)  # (N, K)
    Aq, sa = native.fp8_quantize(A.to(cuda))
    Bq, sb = native.fp8_quantize(B.to(cuda))
    a8, b8 = Aq.view(F8), Bq.view(F8)
    exact = oracle.scaled_mm(Aq.cpu().numpy(), Bq.cpu().numpy(), sa.cpu().numpy(), sb.cpu().numpy())
    bound_q = oracle.abs_dot_bound(Aq.cpu().numpy(), Bq.cpu().numpy(), sa.cpu().numpy(), sb.cpu().numpy())
    for out in (torch._scaled_mm(a8, b8.t(), scale_a=sa, scale_b=sb),
                torch._scaled_mm(a8, b8.t(), sa, sb),
                torch._scaled_mm(Aq, Bq.t(), scale_a=sa, scale_b=sb, out_dtype=torch.float32),
                torch._scaled_mm(a8, b8.t().contiguous(), scale_a=sa, scale_b=sb)):  # row-major `other`: copied
        assert out.shape == (16, 32) and out.dtype == torch.float32 and out.device.type == "cuda"
        assert np.all(np.abs(out.cpu().numpy() - exact) <= 1e-3 * bound_q)  # MFMA accumulation, see test_gpu_parity.MFMA_TOL
    assert oracle.rel_rmse(exact, (A @ B.T).numpy()) < 0.06
    bias = torch.randn(32, generator=g).to(cuda)
    out = torch._scaled_mm(a8, b8.t(), scale_a=sa, scale_b=sb, bias=bias, out_dtype=torch.bfloat16)
    assert out.dtype == torch.bfloat16
    assert np.allclose(out.float().cpu().numpy(), exact + bias.cpu().numpy()[None, :], rtol=2e-2, atol=2e-2)
    out = torch._scaled_mm(a8, b8.t())  # default scales = 1 (fp8_mps_patch.py:87-90)
    exact1 = oracle.scaled_mm(Aq.cpu().numpy(), Bq.cpu().numpy(), [1.0], [1.0], accumulate="f64")
    bound1 = oracle.abs_dot_bound(Aq.cpu().numpy(), Bq.cpu().numpy(), [1.0], [1.0])
    assert np.all(np.abs(out.cpu().numpy() - exact1) <= 1e-3 * bound1)  # MFMA_TOL
    # per-row scales in the (M,1) / (1,N) layout torch >= 2.5 passes
    ra = torch.rand(16, 1, device=cuda) + 0.5
    rb = torch.rand(1, 32, device=cuda) + 0.5
    out = torch._scaled_mm(a8, b8.t(), scale_a=ra, scale_b=rb)
    exp = oracle.scaled_mm(Aq.cpu().numpy(), Bq.cpu().numpy(), ra.cpu().numpy().ravel(), rb.cpu().numpy().ravel(),
                           accumulate="f64")
    bnd = oracle.abs_dot_bound(Aq.cpu().numpy(), Bq.cpu().numpy(), ra.cpu().numpy().ravel(), rb.cpu().numpy().ravel())
    assert np.all(np.abs(out.cpu().numpy() - exp) <= 1e-3 * bnd)  # MFMA_TOL


def test_patched_scaled_mm_any_out_dtype(patch, native, cuda, oracle):
    """The reference ends `_scaled_mm` in `result.to(out_dtype)` whatever the dtype (fp8_mps_patch.py:103-104): a float8_e4m3fn result goes through
    its patched `.to` - the encode kernel with the reference's rounding rules - and float64 through torch."""
    g = torch.Generator().manual_seed(8)
    Aq, sa = native.fp8_quantize(torch.randn(48, 256, generator=g).to(cuda))
    Bq, sb = native.fp8_quantize(torch.randn(64, 256, generator=g).to(cuda))
    sr = torch.tensor([0.37], device=cuda)
    f32 = torch._scaled_mm(Aq.view(F8), Bq.view(F8).t(), scale_a=sa, scale_b=sb, scale_result=sr)
    out = torch._scaled_mm(Aq.view(F8), Bq.view(F8).t(), scale_a=sa, scale_b=sb, scale_result=sr, out_dtype=F8)
    assert out.dtype == F8 and out.shape == (48, 64) and out.device.type == "cuda"
    assert np.array_equal(out.view(torch.uint8).cpu().numpy(), oracle.encode(f32.cpu().numpy()))       # byte-exact: the reference's encoder on the fp32 result
    out = torch._scaled_mm(Aq.view(F8), Bq.view(F8).t(), sa, sb, None, sr, torch.float64)                # positional, torch's order
    assert out.dtype == torch.float64 and torch.equal(out, f32.double())


def test_fp8_linear_call_site_flux_shapes(patch, native, cuda, oracle):
    """A ComfyUI-style fp8 linear (x.to(fp8) -> _scaled_mm(w.t()) -> bf16) at a
    FLUX projection shape, unchanged call site."""
    g = torch.Generator().manual_seed(9)
    M, K, N = 256, 3072, 1536
    w = (torch.randn(N, K, generator=g) * 0.05).to(cuda)
    x = torch.randn(M, K, generator=g).to(cuda)
    w8 = w.to(F8)                                  # weight load: value-preserving encode
    x8 = x.to(F8)
    one = torch.ones((), device=cuda)
    y = torch._scaled_mm(x8, w8.t(), scale_a=one, scale_b=one, out_dtype=torch.bfloat16)
    ref = x @ w.t()
    assert y.shape == (M, N) and y.dtype == torch.bfloat16
    assert oracle.rel_rmse(y.float().cpu().numpy(), ref.cpu().numpy()) < 0.06


def test_fp8_linear_dynamic_quant_chain(native, cuda, oracle):
    """native.fp8_linear = fp8_quantize (amax + encode on the device) -> scaled_mm with fused bias / cast: the bytes
    and the inverse scale equal the oracle's quantize, the product equals the oracle on those bytes, and the result
    is within the reference's ~4 % rel-RMSE of the unquantised fp32 linear (README.md:103-111)."""
    g = torch.Generator().manual_seed(5)
    for (lead, K, N, dt) in (((3, 40), 256, 192, torch.bfloat16), ((64,), 4096, 512, torch.float16), ((2,), 1024, 130, torch.float32)):
        x = torch.randn(*lead, K, generator=g).to(dt)
        W = torch.randn(N, K, generator=g)
        bias = torch.randn(N, generator=g)
        wq, w_inv = oracle.quantize(W.numpy())
        y = native.fp8_linear(x.to(cuda), torch.from_numpy(wq).to(cuda), torch.tensor([w_inv]), bias=bias.to(cuda))
        assert y.shape == (*lead, N) and y.dtype == dt
        x2 = x.float().reshape(-1, K).numpy()
        xq, x_inv = oracle.quantize(x2)
        exact = oracle.scaled_mm(xq, wq, [x_inv], [w_inv], accumulate="f64") + bias.numpy()[None, :].astype(np.float64)
        bound = oracle.abs_dot_bound(xq, wq, [x_inv], [w_inv]) + np.abs(bias.numpy())[None, :]
        got = y.float().cpu().numpy().reshape(-1, N).astype(np.float64)
        eps = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11, torch.float32: 0.0}[dt]
        assert np.all(np.abs(got - exact) <= 1e-3 * bound + eps * np.abs(exact) + 1e-30)
        ref = x2 @ W.numpy().T + bias.numpy()[None, :]
        assert oracle.rel_rmse(got, ref) < 0.06


def test_sharded_linear_on_gpu_single_rank_group(native, cuda, oracle):
    """The N-column-sharded linear through its collective branch (RCCL, side stream, events, in-place gather) with a
    1-rank group - all one GPU allows; world 2 runs under gloo in tests/test_sharded_gloo.py.  The result must equal the
    UNSHARDED fused call bit for bit, bf16 output and bias included (the transposed epilogue adds the bias before the
    single cast and applies the scales in the untransposed order), and a captured HIP graph of the forward replays it."""
    import torch.distributed as dist
    from fp8_sharded_linear import ColumnShardedFP8Linear
    g = torch.Generator().manual_seed(3)
    M, K, N = 384, 1024, 512
    x = torch.randint(0, 120, (M, K), dtype=torch.uint8, generator=g).to(cuda)
    W = torch.randint(0, 120, (N, K), dtype=torch.uint8, generator=g).to(cuda)
    sb = (torch.rand(N, generator=g) * 0.01 + 0.005).to(cuda)
    bias = torch.randn(N, generator=g).to(cuda)
    sa = torch.tensor([0.02], device=cuda)
    exact = oracle.scaled_mm(x.cpu().numpy(), W.cpu().numpy(), [0.02], sb.cpu().numpy(), accumulate="f64") + bias.cpu().numpy()[None, :]
    bound = oracle.abs_dot_bound(x.cpu().numpy(), W.cpu().numpy(), [0.02], sb.cpu().numpy()) + bias.abs().cpu().numpy()[None, :]
    refs = {}
    for od in (torch.float32, torch.bfloat16):
        refs[od] = native.fp8_scaled_mm(x, W, sa, sb, bias=bias, out_dtype=od, split_k=1)     # unsharded, fused
        eps = 2.0 ** -8 if od == torch.bfloat16 else 0.0
        assert np.all(np.abs(refs[od].float().cpu().numpy() - exact) <= 1e-3 * bound + eps * np.abs(exact))
        for chunks in (1, 4):
            lin = ColumnShardedFP8Linear.from_full(W, sb, bias, chunks=chunks, out_dtype=od)   # no group: slots only
            y0 = lin(x, sa)
            assert y0.shape == (M, N) and y0.stride() == (1, M)
            assert torch.equal(y0, refs[od]), (od, chunks)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=cuda)
    try:
        for od in (torch.float32, torch.bfloat16):
            lin = ColumnShardedFP8Linear.from_full(W, sb, bias, chunks=4, out_dtype=od)
            y1 = lin(x, sa)
            torch.cuda.synchronize()
            assert torch.equal(y1, refs[od])
            assert torch.equal(lin(x, sa), refs[od])           # second call: the module's stream / events are reused
        # the forward as a HIP graph (side-stream fork / join inside the capture)
        lin = ColumnShardedFP8Linear.from_full(W, sb, bias, chunks=2, out_dtype=torch.bfloat16)
        lin(x, sa)
        torch.cuda.synchronize()
        side = torch.cuda.Stream(device=cuda)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                yg = lin(x, sa)
        yg.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(yg, refs[torch.bfloat16])
    finally:
        dist.destroy_process_group()


def test_transposed_epilogue_equals_untransposed_bits(native, cuda, oracle):
    """FP8MI_EPILOGUE_TRANSPOSED: C^T = W . X^T with bias per ROW and the scales applied in the untransposed order gives
    the transpose of the plain fused call bit for bit on every tile kernel (same K order in all of them), lands in a
    caller-provided slab with a row stride, and stays within the oracle's bound on the other kernels."""
    import fp8_mi355x_lib as L
    rng = np.random.default_rng(8)
    for (M, K, N) in ((256, 512, 384), (512, 3072, 256), (130, 272, 70)):
        X = torch.from_numpy(rng.integers(0, 127, size=(M, K), dtype=np.uint8)).to(cuda)
        W = torch.from_numpy(rng.integers(0, 127, size=(N, K), dtype=np.uint8)).to(cuda)
        sx = torch.tensor([0.013], device=cuda)
        sw = torch.from_numpy(rng.uniform(0.005, 0.02, size=N).astype(np.float32)).to(cuda)
        b = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(cuda)
        for od in (torch.float32, torch.bfloat16, torch.float16):
            kerns = [L.KERNEL_GEMM_128, L.KERNEL_GEMM_128x64, L.KERNEL_GEMM_256, L.KERNEL_GEMM_64x128]
            if K % 128 == 0 and M % 8 == 0 and N % 8 == 0:   # the one-wave-per-SIMD kernels' envelope (whole K-steps, 16-byte column groups)
                kerns += [L.KERNEL_GEMM_256W, L.KERNEL_GEMM_256x128W]
            for kern in kerns:
                plain = native.fp8_scaled_mm(X, W, sx, sw, bias=b, out_dtype=od, kernel=kern, split_k=1)
                big = torch.full((N + 3, M + 16), 7.0, dtype=od, device=cuda)      # slab with a row stride
                slot = big[2:2 + N, :M]
                got = native.fp8_scaled_mm(W, X, sw, sx, bias=b, out_dtype=od, kernel=kern, split_k=1, out=slot,
                                           transposed_epilogue=True)
                assert got.data_ptr() == slot.data_ptr()
                assert torch.equal(slot.t(), plain), (M, K, N, od, kern)
                assert bool((big[:2] == 7).all()) and bool((big[2 + N:] == 7).all()) and bool((big[:, M:] == 7).all())
    # every other kernel family: the same flag, checked against the oracle (their summation orders differ)
    for (Mx, K, N, kern) in ((1, 512, 64, L.KERNEL_AUTO), (40, 1024, 1, L.KERNEL_AUTO), (40, 1024, 24, L.KERNEL_SKINNY),
                             (9, 100, 7, L.KERNEL_GENERIC)):
        X = rng.integers(0, 127, size=(Mx, K), dtype=np.uint8)
        W = rng.integers(0, 127, size=(N, K), dtype=np.uint8)
        sw = rng.uniform(0.005, 0.02, size=N).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        got = native.fp8_scaled_mm(torch.from_numpy(W).to(cuda), torch.from_numpy(X).to(cuda), torch.from_numpy(sw).to(cuda),
                                   torch.tensor([0.013], device=cuda), bias=torch.from_numpy(b).to(cuda), kernel=kern,
                                   transposed_epilogue=True)
        exact = oracle.scaled_mm(X, W, [0.013], sw, accumulate="f64") + b[None, :]
        bound = oracle.abs_dot_bound(X, W, [0.013], sw) + np.abs(b)[None, :]
        assert got.shape == (N, Mx)
        assert np.all(np.abs(got.cpu().numpy().T - exact) <= 1e-3 * bound + 1e-30), (Mx, K, N, kern)


def _reap(procs):
    """No worker of a multi-process test outlives it (exactly the processes started here): a rank left waiting for a dead peer would keep the GPU busy under
    whatever runs next."""
    for p in procs:
        if p.is_alive():
            p.terminate()
            p.join(timeout=20)
            if p.is_alive():
                p.kill()


def _two_rank_worker(rank, world, port, q):
    """One of two processes sharing the ONE GPU of the box: real HIP kernels for the local product, a gloo group for the gather (RCCL refuses two
    ranks on one device; what is under test is the module's GPU branch - side stream, events, in-place slots - with a peer that really exists)."""
    import sys
    from conftest import PKG
    sys.path.insert(0, PKG)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    try:
        import fp8_mi355x_native as native
        from fp8_sharded_linear import ColumnShardedFP8Linear
        dev = torch.device("cuda:0")
        g = torch.Generator().manual_seed(11)       # the same data on every rank
        M, K, N = 320, 1024, 768
        x = torch.randint(0, 120, (M, K), dtype=torch.uint8, generator=g).to(dev)
        W = torch.randint(0, 120, (N, K), dtype=torch.uint8, generator=g).to(dev)
        sb = (torch.rand(N, generator=g) * 0.01 + 0.005).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        sa = torch.tensor([0.02], device=dev)
        ok = True
        for od in (torch.float32, torch.bfloat16):
            ref = native.fp8_scaled_mm(x, W, sa, sb, bias=bias, out_dtype=od, split_k=1)      # the unsharded fused call
            for chunks in (1, 3):
                lin = ColumnShardedFP8Linear.from_full(W, sb, bias, chunks=chunks, out_dtype=od)
                buf = torch.full((N, M), float("nan"), dtype=od, device=dev)                  # caller-owned gather buffer, poisoned
                for _ in range(2):                                                             # twice: the module's stream / events are reused
                    y = lin(x, sa, out_t=buf)
                    torch.cuda.synchronize()
                    ok = ok and y.data_ptr() == buf.data_ptr() and torch.equal(y, ref)
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_sharded_linear_two_ranks_on_one_gpu(cuda):
    """World 2 with REAL kernels: two processes on the box's one GPU, each multiplying its chunk-cyclic half of the weight rows into its slots of
    the gather buffer, the other half arriving through the group - bit-equal to the unsharded fused call on both ranks (fp32 and bf16, bias and
    per-row scales, 1 and 3 chunks, caller-owned buffer).  The CPU twin with an injected product is tests/test_sharded_gloo.py."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted(q.get(timeout=240) for _ in range(2))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        _reap(procs)
    assert res == [(0, True), (1, True)]


def _peer_gather_worker(rank, world, port, q):
    """One of `world` processes on the box's ONE GPU: the direct all-gather of include/fp8mi_peer.h with peers that really exist (HIP IPC maps
    another process's allocation whichever device it lives on; what a single box cannot show is the xGMI rate)."""
    import sys
    from conftest import PKG
    sys.path.insert(0, PKG)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))   # a rank that died must not leave the others waiting for half an hour
    ok, note = True, ""
    try:
        import fp8_mi355x_native as native
        import fp8_peer_gather
        from fp8_sharded_linear import ColumnShardedFP8Linear
        dev = torch.device("cuda:0")
        # 1. raw slabs: a pattern per (rank, round), slabs of two sizes, 24 rounds back to back without a host sync in between
        slab = 3 * 4096 + 16
        pg = fp8_peer_gather.PeerGather(world * slab, dev, timeout_us=20_000_000)
        buf = pg.tensor(torch.uint8)
        snaps = []
        for it in range(24):
            n = slab if it % 2 == 0 else 4096
            buf[rank * slab:rank * slab + n] = (rank * 37 + it * 11 + torch.arange(n, device=dev)) % 251
            pg.allgather(rank * slab, n)
            snaps.append((it, n, buf.clone()))        # stream-ordered behind the gather: what a consumer would read
        if pg.status() != 0:
            ok, note = False, "status after raw rounds"
        for it, n, snap in snaps:
            for r in range(world):
                want = ((r * 37 + it * 11 + torch.arange(n)) % 251).to(torch.uint8)
                if not torch.equal(snap[r * slab:r * slab + n].cpu(), want):
                    ok, note = False, f"raw round {it} slab of rank {r}"
        # 2. argument errors are errors
        for bad in ((8, 16), (0, 24), (world * slab, 16)):
            try:
                pg.allgather(*bad)
                ok, note = False, f"accepted {bad}"
            except fp8_peer_gather.PeerGatherError:
                pass
        pg.close()
        # 3. a peer that never arrives: bounded waits, status bits instead of a hung GPU
        pg = fp8_peer_gather.PeerGather(4096, dev, timeout_us=300_000)
        if rank == 0:
            pg.allgather(0, 1024)
            st = pg.status()
            if st != (fp8_peer_gather.TIMEOUT_READY | fp8_peer_gather.TIMEOUT_DONE):
                ok, note = False, f"timeout status {st}"
            if pg.status() != 0:
                ok, note = False, "status not cleared"
        pg.close()
        # 4. the sharded linear on it: bit-equal to the unsharded fused call, fresh activations every forward (a stale slab would show)
        g = torch.Generator().manual_seed(13)
        M, K, N = 320, 1024, 768
        W = torch.randint(0, 120, (N, K), dtype=torch.uint8, generator=g).to(dev)
        sb = (torch.rand(N, generator=g) * 0.01 + 0.005).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        sa = torch.tensor([0.02], device=dev)
        for od, chunks in ((torch.float32, 1), (torch.bfloat16, 2)):
            lin = ColumnShardedFP8Linear.from_full(W, sb, bias, chunks=chunks, out_dtype=od, gather="peer", max_tokens=M)
            for it in range(6):
                Mi = M if it % 3 else M - 64                      # fewer tokens than the buffer was made for
                x = torch.randint(0, 120, (Mi, K), dtype=torch.uint8, generator=g).to(dev)
                y = lin(x, sa)
                ref = native.fp8_scaled_mm(x, W, sa, sb, bias=bias, out_dtype=od, split_k=1)
                if not torch.equal(y, ref):
                    ok, note = False, f"sharded linear {od} chunks {chunks} forward {it}"
            if lin._peer.status() != 0:
                ok, note = False, "status after the sharded linear"
            lin.close()
        # 5. two layers through ONE mapped buffer (what bench.py's sharded workload does with its rotating weight buffers): the second layer's slabs
        #    must not land before the first layer's result has been consumed on every rank
        shared = fp8_peer_gather.PeerGather(N * M * 4, dev, timeout_us=20_000_000)
        W2 = torch.randint(0, 120, (N, K), dtype=torch.uint8, generator=g).to(dev)
        lins = [ColumnShardedFP8Linear.from_full(w_, sb, bias, chunks=2, out_dtype=torch.float32, peer=shared) for w_ in (W, W2)]
        x = torch.randint(0, 120, (M, K), dtype=torch.uint8, generator=g).to(dev)
        outs = [lins[i % 2](x, sa).clone() for i in range(6)]          # (the clone is the consumer: stream-ordered behind the gather)
        refs = [native.fp8_scaled_mm(x, w_, sa, sb, bias=bias, out_dtype=torch.float32, split_k=1) for w_ in (W, W2)]
        if not all(torch.equal(o, refs[i % 2]) for i, o in enumerate(outs)) or shared.status() != 0:
            ok, note = False, "two layers sharing one peer buffer"
        for l_ in lins:
            l_.close()                                                   # (does not own the buffer)
        shared.close()
    except Exception as e:        # noqa: BLE001 - reported through the queue; the parent asserts
        ok, note = False, f"{type(e).__name__}: {e}"
    finally:
        q.put((rank, ok, note))
        dist.destroy_process_group()


@pytest.mark.gpu
def test_peer_allgather_three_ranks_on_one_gpu(cuda):
    """include/fp8mi_peer.h end to end with three processes sharing the box's GPU: IPC export / open, the ready -> push -> done protocol over 24
    unsynchronised rounds, argument errors, the bounded wait (a rank that never calls: status bits, no hang) and ColumnShardedFP8Linear(gather=
    'peer') bit-equal to the unsharded call."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 3
    procs = [ctx.Process(target=_peer_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted(q.get(timeout=300) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        _reap(procs)
    assert res == [(r, True, "") for r in range(world)], res
