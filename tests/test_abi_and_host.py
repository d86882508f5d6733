"""CPU: the C-ABI library loads and exports every symbol include/fp8mi.h
declares (no compute without a GPU), and the host-side routing logic of the
monkey-patch behaves like the reference's (fp8_mps_patch.py)."""
import ctypes
import os
import re
import subprocess

import pytest
import torch

from conftest import PKG, ROOT


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(PKG, "libfp8mi.so")
    srcs = [os.path.join(PKG, "csrc", f) for f in os.listdir(os.path.join(PKG, "csrc")) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(ROOT, "include", "fp8mi.h"))
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", PKG, "-s", "-j4"])
    import fp8_mi355x_lib
    return fp8_mi355x_lib.load()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "fp8mi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fp8mi_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    import fp8_mi355x_lib
    names = _declared_symbols()
    assert {"fp8mi_scaled_mm", "fp8mi_scaled_mm_ex", "fp8mi_dequant", "fp8mi_encode", "fp8mi_amax",
            "fp8mi_quantize", "fp8mi_device_info", "fp8mi_version", "fp8mi_last_error",
            "fp8mi_profile_begin", "fp8mi_profile_end"} <= set(names)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/fp8mi.h but not exported"
        assert n in fp8_mi355x_lib.SIGNATURES, f"{n} has no ctypes signature"
    assert set(fp8_mi355x_lib.SIGNATURES) == set(names)


def test_version_and_argument_errors_without_gpu(lib):
    assert lib.fp8mi_version() == 0x000400
    # argument validation happens before any HIP call, so it is testable here
    rc = lib.fp8mi_scaled_mm(None, None, None, None, None, None, None, 4, 4, 4, 4, 4, 4, 0, 0, 0, 0, 0, None)
    assert rc == -1 and b"NULL" in lib.fp8mi_last_error()
    rc = lib.fp8mi_scaled_mm(None, None, None, None, None, None, None, -1, 4, 4, 4, 4, 4, 0, 0, 0, 0, 0, None)
    assert rc == -2
    one = ctypes.c_void_p(16)
    rc = lib.fp8mi_scaled_mm(one, one, one, one, one, None, None, 4, 4, 8, 4, 8, 4, 0, 0, 0, 0, 0, None)
    assert rc == -2 and b"leading dimension" in lib.fp8mi_last_error()
    rc = lib.fp8mi_scaled_mm(one, one, one, one, one, None, None, 4, 4, 8, 8, 8, 4, 0, 0, 7, 0, 0, None)
    assert rc == -3
    assert lib.fp8mi_dequant(None, None, None, 0, 0, None) == 0      # empty is a no-op
    assert lib.fp8mi_dequant(None, None, None, 5, 0, None) == -1
    assert lib.fp8mi_encode(one, 9, one, None, 5, 0, None) == -3
    assert lib.fp8mi_scaled_mm(None, None, None, None, None, None, None, 0, 4, 4, 4, 4, 4, 0, 0, 0, 0, 0, None) == 0
    # split-K entry point: same validation; split_k < 0 is an enum error; the workspace size is a constant
    assert lib.fp8mi_scaled_mm_ws(one, one, one, one, one, None, None, -1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, None, 0, None) == -2
    rc = lib.fp8mi_scaled_mm_ws(one, one, one, one, one, None, None, 4, 4, 16, 16, 16, 4, 0, 0, 0, 0, 0, 0, -1, None, 0, None)
    assert rc == -3 and b"split_k" in lib.fp8mi_last_error()
    assert lib.fp8mi_scaled_mm_workspace_bytes() >= 4096 + 256 * 128 * 64 * 4


def test_peer_library_exports_what_its_header_declares_and_validates_arguments():
    """include/fp8mi_peer.h (the direct all-gather): every declared entry point exported and bound by the Python host; the argument checks
    that precede any HIP call.  The data path needs peers: tests/test_gpu_patch.py::test_peer_allgather_three_ranks_on_one_gpu."""
    so = os.path.join(PKG, "libfp8mi_peer.so")
    srcs = [os.path.join(PKG, "csrc", "peer", "fp8mi_peer.hip"), os.path.join(ROOT, "include", "fp8mi_peer.h")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", PKG, "-s", "libfp8mi_peer.so"])
    import fp8_peer_gather
    lib = fp8_peer_gather.load()
    text = re.sub(r"/\*.*?\*/", "", open(srcs[1]).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(fp8mi_peer_[a-z0-9_]+)\s*\(", text)))
    assert len(names) == 11 and "fp8mi_peer_allgather" in names
    for n in names:
        fn = getattr(lib, n)                                   # AttributeError = declared but not exported
        assert fn.restype is not None, n
        assert n in ("fp8mi_peer_version", "fp8mi_peer_last_error") or fn.argtypes is not None, f"{n} has no ctypes signature"
    assert lib.fp8mi_peer_version() == 0x000100
    vp = ctypes.c_void_p
    two = (vp * 2)(16, 32)
    ctx = vp()
    assert lib.fp8mi_peer_ctx_create(1, 0, two, two, 4096, ctypes.byref(ctx)) == -2 and b"world" in lib.fp8mi_peer_last_error()
    assert lib.fp8mi_peer_ctx_create(2, 2, two, two, 4096, ctypes.byref(ctx)) == -2
    assert lib.fp8mi_peer_ctx_create(2, 0, two, two, 4100, ctypes.byref(ctx)) == -2 and b"multiple of 16" in lib.fp8mi_peer_last_error()
    assert lib.fp8mi_peer_ctx_create(2, 0, (vp * 2)(16, 0), two, 4096, ctypes.byref(ctx)) == -1
    assert lib.fp8mi_peer_ctx_create(2, 0, None, two, 4096, ctypes.byref(ctx)) == -1
    assert lib.fp8mi_peer_allgather(None, 0, 16, 0, None) == -1
    assert lib.fp8mi_peer_alloc(16, 0, None) == -1 and lib.fp8mi_peer_export(None, None) == -1 and lib.fp8mi_peer_open(None, None) == -1
    assert lib.fp8mi_peer_free(None) == 0 and lib.fp8mi_peer_close(None) == 0


def test_peer_gather_host_refuses_what_it_cannot_do(monkeypatch):
    import fp8_peer_gather
    from fp8_sharded_linear import ColumnShardedFP8Linear
    with pytest.raises(fp8_peer_gather.PeerGatherError, match="process group"):
        fp8_peer_gather.PeerGather(4096, torch.device("cuda:0"))
    w = torch.zeros(8, 16, dtype=torch.uint8)
    with pytest.raises(ValueError, match="at least 2 ranks"):
        ColumnShardedFP8Linear(w, torch.ones(1), N=8, gather="peer")
    with pytest.raises(ValueError, match="'rccl' or 'peer'"):
        ColumnShardedFP8Linear(w, torch.ones(1), N=8, gather="ring")
    monkeypatch.setattr(fp8_peer_gather, "_lib", None)
    monkeypatch.setattr(fp8_peer_gather.os.path, "exists", lambda p: False)
    with pytest.raises(ImportError, match="libfp8mi_peer.so"):
        fp8_peer_gather.load()


def test_missing_library_fails_loudly(monkeypatch):
    import fp8_mi355x_lib
    monkeypatch.setattr(fp8_mi355x_lib, "_lib", None)
    monkeypatch.setattr(fp8_mi355x_lib, "LIB_PATH", "/nonexistent/libfp8mi.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        fp8_mi355x_lib.load()


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(root, f)).read()
                assert "fp8_oracle" not in text and "oracle/" not in text, f


# ---- monkey-patch surface (test_fp8_metal.py:318-349, test_mps_limits_patch.py:128-153) ----

def test_patch_module_structure():
    import fp8_mps_patch as p
    for name in ("install", "uninstall", "is_installed", "patch_vae_decode_for_mps_limits",
                 "_metal_scaled_mm", "_metal_tensor_to", "_metal_tensor_copy"):
        assert callable(getattr(p, name)), name


def test_install_uninstall_restores_identical_objects():
    import fp8_mps_patch as p
    o_mm, o_to, o_cp = torch._scaled_mm, torch.Tensor.to, torch.Tensor.copy_
    assert not p.is_installed()
    p.install()
    try:
        assert p.is_installed()
        assert torch._scaled_mm is p._metal_scaled_mm
        assert torch.Tensor.to is p._metal_tensor_to and torch.Tensor.copy_ is p._metal_tensor_copy
        assert p._original_scaled_mm is o_mm and p._original_tensor_to is o_to and p._original_tensor_copy is o_cp
        p.install()  # idempotent
        assert p._original_scaled_mm is o_mm
    finally:
        p.uninstall()
    assert not p.is_installed()
    assert torch._scaled_mm is o_mm and torch.Tensor.to is o_to and torch.Tensor.copy_ is o_cp
    assert p._original_scaled_mm is None and p._original_tensor_to is None and p._original_tensor_copy is None
    p.uninstall()  # no-op


def test_cpu_calls_pass_through_unchanged(patch):
    """With the patch installed, CPU tensors behave exactly as unpatched."""
    x = torch.randn(4, 8)
    assert x.to(torch.float16).dtype == torch.float16
    assert x.to("cpu", torch.float64).dtype == torch.float64
    ref = patch._original_tensor_to(x, torch.float8_e4m3fn)
    got = x.to(torch.float8_e4m3fn)
    assert got.dtype == torch.float8_e4m3fn and torch.equal(got.view(torch.uint8), ref.view(torch.uint8))
    assert torch.equal(got.to(torch.float32), patch._original_tensor_to(ref, torch.float32))
    y = torch.empty(4, 8, dtype=torch.float8_e4m3fn)
    assert y.copy_(x) is y and torch.equal(y.view(torch.uint8), ref.view(torch.uint8))
    z = torch.zeros(4, 8)
    z.copy_(x)
    assert torch.equal(z, x)
    z.copy_(torch.tensor(3.0))
    assert torch.all(z == 3)
    assert x.to(x.to(torch.float16)).dtype == torch.float16  # to(other) overload


def test_to_argument_parsing():
    import fp8_mps_patch as p
    f8 = torch.float8_e4m3fn
    assert p._parse_to_args((f8,), {}) == (f8, None, {})
    assert p._parse_to_args(("cuda",), {}) == (None, "cuda", {})
    assert p._parse_to_args(("cuda:1", f8), {}) == (f8, "cuda:1", {})
    assert p._parse_to_args((torch.device("cuda", 0), f8, True), {}) == (f8, torch.device("cuda", 0), {"non_blocking": True})
    assert p._parse_to_args((), {"dtype": f8, "device": "cuda", "copy": True}) == (f8, "cuda", {"copy": True})
    d, dev, extra = p._parse_to_args((), {"dtype": torch.float16, "memory_format": torch.contiguous_format})
    assert d == torch.float16 and dev is None and extra == {"memory_format": torch.contiguous_format}
    t = torch.zeros(1, dtype=torch.float16)
    assert p._parse_to_args((t,), {})[:2] == (torch.float16, t.device)


def test_to_routing_table():
    """The scenario routing of _metal_tensor_to (fp8_mps_patch.py:160-226),
    with "cuda" in the role of "mps"."""
    import fp8_mps_patch as p
    f8, f5 = torch.float8_e4m3fn, torch.float8_e5m2
    S = p._to_scenario
    # S1: fp8 elsewhere -> device: raw byte transfer
    assert S(f8, False, None, "cuda") == "bytes_to_device"
    assert S(f8, False, f8, "cuda:0") == "bytes_to_device"
    assert S(f5, False, None, "cuda") == "bytes_to_device"
    assert S(f8, False, torch.float16, "cuda") == "bytes_to_device"   # then dequantised
    assert S(f8, False, f5, "cuda") == "original"
    # S2: float -> e4m3 on device: encode kernel
    assert S(torch.float32, True, f8, None) == "encode"
    assert S(torch.bfloat16, False, f8, "cuda") == "encode"
    assert S(torch.float32, False, f8, None) == "original"             # CPU stays CPU
    assert S(torch.float32, True, f5, None) == "original"              # e5m2 is torch's business
    # S3: e4m3 on device
    assert S(f8, True, None, None) == "same" and S(f8, True, f8, "cuda") == "same"
    assert S(f8, True, torch.float16, None) == "dequant" and S(f8, True, torch.float32, "cuda") == "dequant"
    assert S(f8, True, f5, None) == "original"
    assert S(f8, True, torch.int32, None) == "original"
    assert S(f8, True, None, "cpu") == "original" and S(f5, True, torch.float16, None) == "original"
    # nothing fp8: always torch
    assert S(torch.float32, True, torch.float16, None) == "original"
    assert S(torch.float32, False, None, "cuda") == "original"


def test_copy_routing_table():
    import fp8_mps_patch as p
    f8, f5 = torch.float8_e4m3fn, torch.float8_e5m2
    C = p._copy_scenario
    assert C(f8, True, f8) == "bytes" and C(f5, True, f5) == "bytes"
    assert C(f8, True, torch.float32) == "encode" and C(f8, True, torch.bfloat16) == "encode"
    assert C(f8, True, f5) == "original" and C(f5, True, torch.float32) == "original"
    assert C(torch.float32, True, f8) == "original"                    # fp8_mps_patch.py:296-299
    assert C(f8, False, torch.float32) == "original" and C(f8, False, f8) == "original"


def test_scaled_mm_cpu_passthrough_and_positional_scales(patch):
    a = torch.randn(16, 32).to(torch.float8_e4m3fn)
    b = torch.randn(16, 32).to(torch.float8_e4m3fn).t()
    one = torch.ones(1)
    try:
        ref = patch._original_scaled_mm(a, b, scale_a=one, scale_b=one, out_dtype=torch.float32)
    except Exception as e:
        pytest.skip(f"torch-CPU _scaled_mm unavailable: {e}")
    assert torch.equal(torch._scaled_mm(a, b, scale_a=one, scale_b=one, out_dtype=torch.float32), ref)
    assert torch.equal(torch._scaled_mm(a, b, one, one, out_dtype=torch.float32), ref)


def test_plugin_entry_installs_on_import_and_exports_empty_node_maps(capsys):
    """The ComfyUI custom-node entry (reference __init__.py:13-61): importing the package directory's
    __init__.py installs the patch as a side effect (:22-27), prints a banner, and exports empty node
    mappings (:57-61).  ComfyUI loads custom nodes with importlib from the file path, so do the same."""
    import importlib.util
    import sys
    import fp8_mps_patch as p
    assert not p.is_installed()
    o_mm, o_to, o_cp = torch._scaled_mm, torch.Tensor.to, torch.Tensor.copy_
    spec = importlib.util.spec_from_file_location("fp8_mi355x_plugin_cpu_test", os.path.join(PKG, "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    try:
        spec.loader.exec_module(mod)
        assert p.is_installed() and torch._scaled_mm is p._metal_scaled_mm
        assert mod.NODE_CLASS_MAPPINGS == {} and mod.NODE_DISPLAY_NAME_MAPPINGS == {}
        assert sorted(mod.__all__) == ["NODE_CLASS_MAPPINGS", "NODE_DISPLAY_NAME_MAPPINGS"]
        assert "patch installed" in capsys.readouterr().out
        spec.loader.exec_module(mod)        # a second import leaves one installation (install() is idempotent)
        assert p._original_scaled_mm is o_mm
    finally:
        p.uninstall()
        sys.modules.pop(spec.name, None)
    assert torch._scaled_mm is o_mm and torch.Tensor.to is o_to and torch.Tensor.copy_ is o_cp


def test_plugin_entry_swallows_install_failure(monkeypatch, capsys):
    """reference __init__.py:43-53: a failing install() is reported, not raised, so the host still starts."""
    import importlib.util
    import fp8_mps_patch as p

    def boom():
        raise RuntimeError("torch._scaled_mm not found")
    monkeypatch.setattr(p, "install", boom)
    spec = importlib.util.spec_from_file_location("fp8_mi355x_plugin_fail_test", os.path.join(PKG, "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert not p.is_installed() and mod.NODE_CLASS_MAPPINGS == {}
    assert "WARNING" in capsys.readouterr().out


def test_product_kernels_do_not_spill():
    """Compiles the product kernels with -save-temps and reads the register metadata (tools/check_spills.py): no kernel may
    touch scratch (round 1 shipped the 256x256 GEMM with a spilled VGPR).  ~30 s of hipcc."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_spills.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]


def test_generated_gemm_loop_is_current():
    """csrc/fp8mi_gemm256_loop.inc is generated (csrc/gen/gen_gemm256_loop.py) and committed: the committed text must be
    what the generator produces now (an edited generator without a regenerated loop would ship a stale schedule)."""
    import sys
    gen = os.path.join(ROOT, "fp8-mps-metal_amd", "csrc", "gen", "gen_gemm256_loop.py")
    inc = os.path.join(ROOT, "fp8-mps-metal_amd", "csrc", "fp8mi_gemm256_loop.inc")
    r = subprocess.run([sys.executable, gen], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout == open(inc).read(), "regenerate: python csrc/gen/gen_gemm256_loop.py > csrc/fp8mi_gemm256_loop.inc"


def _dispatch_lib():
    import sys
    from conftest import PKG
    sys.path.insert(0, PKG)
    import fp8_mi355x_lib as L
    return L, L.load()


def test_automatic_dispatch_table():
    """fp8mi_choose_kernel (host-only: shapes, strides, alignment, CU count - 256 when no device is visible, as on the MI355X): the hard RULES
    of the dispatch and the BASELINE configurations.  M = 1 is the reference's own rule (fp8_mps_native.py:193-210); everything else is the
    cheapest kernel by the cost model of csrc/fp8mi_dispatch.h, whose quality is pinned by the next test against measured times."""
    L, lib = _dispatch_lib()

    def pick(M, K, N, out=L.BF16, ws=1, split=0, lda=None, ldb=None):
        return lib.fp8mi_choose_kernel(M, N, K, lda or K, ldb or K, N, out, ws, split)

    assert pick(1, 4096, 4096) == L.KERNEL_GEMV and pick(1, 14336, 4096) == L.KERNEL_GEMV and pick(1, 1024, 13824) == L.KERNEL_GEMV   # configs C1, C2: M = 1 is a rule
    assert pick(512, 4096, 4096, out=L.F32) == L.KERNEL_GEMM_128x64                                # config C3
    assert pick(4096, 3072, 12288) == L.KERNEL_GEMM_256W and pick(8192, 8192, 8192) == L.KERNEL_GEMM_256W   # config C4 (FLUX), 8192^3
    assert pick(4096, 3072, 1536) == L.KERNEL_GEMM_256x128W and pick(1536, 3072, 4096) == L.KERNEL_GEMM_256x128W   # C4's per-rank shard and its transposed form
    assert pick(2048, 4096, 4096) == L.KERNEL_GEMM_256x128W and pick(1024, 4096, 4096, out=L.F32) == L.KERNEL_GEMM_128D   # bench.py's `mid` and `wide`
    assert pick(4, 4096, 4096, out=L.F32) == L.KERNEL_GEMV_MX and pick(64, 14336, 4096) == L.KERNEL_GEMM_64x64              # ... `skinny` (the reference's batch-4 shape) and `decode`
    assert pick(4173, 3072, 12296) == L.KERNEL_GEMM_256W                                           # ragged M and N stay on the one-wave-per-SIMD kernel
    assert pick(4096, 3088, 12288) == L.KERNEL_GEMM_256W                                           # K tail: staged with per-lane masks
    assert pick(4096, 3072, 12292) == L.KERNEL_GEMM_256                                            # N not a multiple of 8 half columns: the ring kernel stands in
    # envelopes: what no tuned kernel reads runs on the generic kernel
    assert pick(5, 100, 7) == L.KERNEL_GENERIC                                                     # K % 16 != 0
    assert pick(512, 4096, 4096, lda=4100) == L.KERNEL_GENERIC                                     # rows not 16-byte aligned
    assert pick(4, 0, 16384, lda=16, ldb=16) == L.KERNEL_GENERIC and pick(512, 0, 4096, lda=16, ldb=16) == L.KERNEL_GENERIC   # K = 0: the empty sum (bias only), never a tile kernel
    assert lib.fp8mi_choose_kernel(-1, 1, 1, 1, 1, 1, 0, 0, 0) < 0
    # every id the dispatch returns is a kernel the model priced (i.e. one that takes the problem), with or without a workspace, forced splits included
    priced = [L.KERNEL_GEMV_MX, L.KERNEL_SKINNY, L.KERNEL_GEMM_32x32, L.KERNEL_GEMM_32x64, L.KERNEL_GEMM_64x64, L.KERNEL_GEMM_64x128, L.KERNEL_GEMM_128x64,
              L.KERNEL_GEMM_128, L.KERNEL_GEMM_128D, L.KERNEL_GEMM_256x128W, L.KERNEL_GEMM_256W, L.KERNEL_GEMM_256]
    for M in (2, 3, 8, 9, 33, 64, 65, 129, 192, 193, 257, 1000, 5000):
        for (K, N) in ((16, 8), (128, 4096), (272, 72), (4096, 4096), (14336, 4096), (1024, 28672), (28672, 1024)):
            for ws, split in ((1, 0), (0, 0), (1, 1), (1, 3)):
                k = pick(M, K, N, ws=ws, split=split)
                assert k in priced, (M, K, N, ws, split, k)
                us = lib.fp8mi_predict_kernel_us(k, M, N, K, K, K, N, L.BF16, ws, split, 0)
                floor = 4.0   # (no dispatch takes less: below it the candidates tie and the first one in the list runs)
                assert us > 0 and all(not (0 <= max(lib.fp8mi_predict_kernel_us(o, M, N, K, K, K, N, L.BF16, ws, split, 0), floor) < max(us, floor)) for o in priced
                                      if lib.fp8mi_predict_kernel_us(o, M, N, K, K, K, N, L.BF16, ws, split, 0) >= 0), (M, K, N)
    # a 32-CU partition (one CPX partition = one XCD) is DEFINED behaviour: finite prices, and fewer slots per round move the choice to larger tiles
    us32 = {k: lib.fp8mi_predict_kernel_us(k, 512, 4096, 4096, 4096, 4096, 4096, L.F32, 1, 0, 32) for k in priced}
    us256 = {k: lib.fp8mi_predict_kernel_us(k, 512, 4096, 4096, 4096, 4096, 4096, L.F32, 1, 0, 256) for k in priced}
    assert min((v, k) for k, v in us256.items() if v > 0)[1] == L.KERNEL_GEMM_128x64
    assert min((v, k) for k, v in us32.items() if v > 0)[1] in (L.KERNEL_GEMM_256W, L.KERNEL_GEMM_256x128W)
    assert all(us32[k] > us256[k] for k in priced if us256[k] > 0)


@pytest.mark.parametrize("fixture,ws,shapes,max_110,max_120", [
    # (bounds = what the shipped constants score, plus a little slack; in brackets what the hand-written rules of round 3 score on the same data:
    #  tools/dispatch_fit/compare.py with a library built from 5637f2f)
    ("dispatch_times_cold_fit.json", 1, 3843, 112, 26),        # fitted on: standard, small / ragged and 200 <= M <= 1024 sweeps, fp32 output   [rules: 240 / 101 of 3,838]
    ("dispatch_times_cold_anchors.json", 1, 28, 1, 0),          # the BASELINE configs, bench.py's workloads and their neighbours; fitted on     [rules: 0 / 0]
    ("dispatch_times_cold_heldout.json", 1, 761, 22, 3),         # NEVER fitted on                                                               [rules: 37 / 15 of 761]
    ("dispatch_times_cold_nows.json", 0, 424, 12, 4)])           # no split-K workspace (a sharded linear's calls); never fitted on               [rules: 137 / 112 of 413]
def test_dispatch_cost_model_against_measured_times(golden_dir, fixture, ws, shapes, max_110, max_120):
    """The cost model's choices against MEASURED times (tests/golden/dispatch_times_cold_*.json: every product kernel that takes a shape, timed on MI355X by
    tools/sweep_regret.py with cold weights, round 4).  Regret = time of the kernel the dispatch picks / time of the fastest.  A change to the model or to its constants
    (tools/dispatch_fit/) must not do worse than the bounds, which are what the shipped constants score."""
    import json
    import os
    import statistics
    L, lib = _dispatch_lib()
    ids = {"mx": L.KERNEL_GEMV_MX, "skinny": L.KERNEL_SKINNY, "32x32": L.KERNEL_GEMM_32x32, "32x64": L.KERNEL_GEMM_32x64, "64x64": L.KERNEL_GEMM_64x64,
           "64x128": L.KERNEL_GEMM_64x128, "128x64": L.KERNEL_GEMM_128x64, "128": L.KERNEL_GEMM_128, "128D": L.KERNEL_GEMM_128D,
           "256W": L.KERNEL_GEMM_256W, "256x128W": L.KERNEL_GEMM_256x128W, "gemv": L.KERNEL_GEMV}
    name = {v: k for k, v in ids.items()}
    doc = json.load(open(os.path.join(golden_dir, fixture)))
    regrets, errs = [], []
    for M, K, N, out, times in doc["shapes"]:
        oc = L.F32 if out == "f32" else L.BF16
        k = lib.fp8mi_choose_kernel(M, N, K, K, K, N, oc, ws, 0)
        if k == L.KERNEL_GEMM_256:      # the 8-wave ring kernel stands in below the one-wave-per-SIMD kernels' envelope (K < 256): the sweeps did not time it
            continue
        if name.get(k) not in times:
            assert False, (M, K, N, out, k)   # the pick is a kernel the sweep did not time on the shape: the model offers it outside its measured range
        # the pick is one of the kernels that was measured on the shape
        regrets.append(times[name[k]] / min(times.values()))
        for kn, t in times.items():
            if kn != "gemv" and M > 1:   # (M = 1 is a rule: the tile kernels measured there are not offered)
                us = lib.fp8mi_predict_kernel_us(ids[kn], M, N, K, K, K, N, oc, ws, 0, 256)
                assert us > 0, (M, K, N, kn)
                errs.append(abs(us / t - 1.0))
    assert shapes - 2 <= len(regrets) <= shapes
    assert statistics.median(regrets) <= 1.005
    assert sum(r > 1.10 for r in regrets) <= max_110 and sum(r > 1.20 for r in regrets) <= max_120, (sum(r > 1.10 for r in regrets), sum(r > 1.20 for r in regrets))
    assert statistics.median(errs) <= (0.08 if ws else 0.12)   # the prices themselves: median |predicted / measured - 1| over every (shape, kernel) pair


def test_bench_probe_libraries_link_and_load():
    """bench.py's measurement probes (tools/libceiling_probe.so, tools/libfloor_probe.so; built by __graft_entry__.build()) load with every symbol resolved
    and export what bench.py binds.  The floor probe compiles the PRODUCT's ring-kernel source a second time, so a new dependency of that source
    (round 4: the dispatch's cost model asks the other kernels' envelopes) must be met there too - an unresolved symbol only shows at load time."""
    import ctypes
    import subprocess
    for so, src, deps, syms, flags in (("libceiling_probe.so", "ceiling_probe.hip", [], ["probe_mfma", "probe_read", "probe_abi_version"], []),
                                       ("libfloor_probe.so", "floor_probe.hip", ["fp8-mps-metal_amd/csrc/fp8mi_gemm.hip", "fp8-mps-metal_amd/csrc/fp8mi_dispatch.h",
                                                                                 "fp8-mps-metal_amd/csrc/fp8mi_gemm_epi.h"], ["floor_probe_run", "floor_probe_abi_version"],
                                        ["-fvisibility=hidden", "-std=c++17"])):
        path, source = os.path.join(ROOT, "tools", so), os.path.join(ROOT, "tools", src)
        newest = max(os.path.getmtime(f) for f in [source] + [os.path.join(ROOT, d) for d in deps])
        if not os.path.exists(path) or os.path.getmtime(path) < newest:
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared"] + flags + [source, "-o", path])
        lib = ctypes.CDLL(path, mode=os.RTLD_NOW)   # RTLD_NOW: every symbol is resolved here, not at the first call
        for sname in syms:
            assert hasattr(lib, sname), (so, sname)


def test_pad_weight_rows_host_logic():
    """native.pad_weight_rows is plain tensor bookkeeping (no kernel): values and dtype kept, row stride K + pad, 16-byte row
    alignment enforced, pad 0 returns the tensor itself."""
    import sys
    from conftest import PKG
    sys.path.insert(0, PKG)
    import fp8_mi355x_native as native
    w = torch.randint(0, 255, (6, 8192), dtype=torch.uint8)
    v = native.pad_weight_rows(w)
    assert v.shape == w.shape and v.stride() == (8192 + 256, 1) and torch.equal(v, w)
    v8 = native.pad_weight_rows(w.view(torch.float8_e4m3fn), 512)
    assert v8.dtype == torch.float8_e4m3fn and v8.stride() == (8192 + 512, 1) and torch.equal(v8.view(torch.uint8), w)
    assert v8.t().stride() == (1, 8192 + 512)                                  # what the patched torch._scaled_mm receives as `other`
    assert native.pad_weight_rows(w, 0) is w
    with pytest.raises(AssertionError):
        native.pad_weight_rows(w, 100)
    with pytest.raises(AssertionError):
        native.pad_weight_rows(torch.zeros(4, 8192))
