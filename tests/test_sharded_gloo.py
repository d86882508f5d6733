"""CPU, world_size 2 over gloo: the N-column-sharded linear + all-gather
(fp8_sharded_linear.py) reproduces the unsharded product.  The local
product is injected (the oracle) because the product itself has no CPU path -
what is under test is the sharding, the transposed-block layout and the gather."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ORACLE, PKG


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _acc_f32(a_u8, b_u8):
    """float32(sum_k dec(a) dec(b)) rounded ONCE from the exact float64 sum: what any summation order converges to,
    so the sharded and the unsharded call see the same accumulator bits (as the GPU kernels do: every tile kernel adds
    the K-steps of an element in the same order)."""
    import fp8_oracle as o
    return (o.decode(a_u8.numpy()).astype(np.float64) @ o.decode(b_u8.numpy()).astype(np.float64).T).astype(np.float32)


def _fused_mm_transposed(w_rows, x_u8, scale_w, scale_x, bias, out):
    """CPU stand-in for libfp8mi's transposed-epilogue call (FP8MI_EPILOGUE_TRANSPOSED): out (rows, M) =
    cast(((acc * s_x) * s_w[row]) + bias[row]) in float32 arithmetic, the order of the untransposed fused epilogue."""
    acc = _acc_f32(w_rows, x_u8)
    sw = scale_w.numpy().reshape(-1).astype(np.float32)
    sx = scale_x.numpy().reshape(-1).astype(np.float32)
    r = (acc * sx[0]) * (sw[:, None] if sw.size > 1 else sw[0])
    if bias is not None:
        r = r + bias.float().numpy()[:, None]
    out.copy_(torch.from_numpy(r.astype(np.float32)).to(out.dtype))
    return out


def _fused_mm_reference(x_u8, W_u8, sa, sb, bias, out_dtype):
    """The UNSHARDED fused epilogue (fp8_matmul.metal:144-146 then fp8_mps_patch.py:94-104): (acc * sa) * sb[n] + bias[n], cast."""
    acc = _acc_f32(x_u8, W_u8)
    sbv = sb.numpy().reshape(-1).astype(np.float32)
    r = (acc * np.float32(sa.numpy().reshape(-1)[0])) * (sbv[None, :] if sbv.size > 1 else sbv[0])
    if bias is not None:
        r = r + bias.float().numpy()[None, :]
    return torch.from_numpy(r.astype(np.float32)).to(out_dtype)


def _worker(rank, world, port, chunks, per_row, with_bias, out_dtype, q, geometry=None):
    for p in (PKG, ORACLE):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fp8_oracle as o
        from fp8_sharded_linear import ColumnShardedFP8Linear, shard_rows
        rng = np.random.default_rng(42)  # same data on every rank
        M, K, N = geometry or (24, 64, 32 * world)
        x = torch.from_numpy(rng.integers(0, 256, size=(M, K), dtype=np.uint8))
        W = torch.from_numpy(rng.integers(0, 256, size=(N, K), dtype=np.uint8))
        sa = torch.tensor([0.03])
        sb = torch.from_numpy(rng.uniform(0.01, 0.05, size=N).astype(np.float32)) if per_row else torch.tensor([0.02])
        bias = torch.from_numpy(rng.standard_normal(N).astype(np.float32)) if with_bias else None
        lin = ColumnShardedFP8Linear.from_full(W, sb, bias, chunks=chunks, out_dtype=out_dtype, mm=_fused_mm_transposed)
        # every rank owns N/world rows, all ranks together cover each row exactly once
        rows = [shard_rows(N, world, r, chunks) for r in range(world)]
        assert sorted(torch.cat(rows).tolist()) == list(range(N))
        y = lin(x, sa)
        assert y.shape == (M, N) and y.stride() == (1, M) and y.dtype == out_dtype  # .t() view of the gathered C^T
        # bit equality with the unsharded fused result (bias added BEFORE the single cast, scales in the same order)
        ok = torch.equal(y, _fused_mm_reference(x, W, sa, sb, bias, out_dtype))
        # ... which itself is the oracle's product
        ref = o.scaled_mm(x.numpy(), W.numpy(), sa.numpy(), sb.numpy(), accumulate="f64")
        bound = o.abs_dot_bound(x.numpy(), W.numpy(), sa.numpy(), sb.numpy())
        if bias is not None:
            ref = ref + bias.numpy()[None, :]
            bound = bound + np.abs(bias.numpy())[None, :]
        eps = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11, torch.float32: 0.0}[out_dtype]
        ok = ok and bool(np.all(np.abs(y.float().numpy() - ref) <= 4e-6 * bound + eps * np.abs(ref) + 1e-30))
        # a second forward reuses the module's state (streams / events / views) and gives the same bits
        ok = ok and torch.equal(lin(x, sa), y)
        # ... and so does one into a caller-owned gather buffer (no allocation in the forward), whose storage it returns
        buf = torch.full((N, M), float("nan"), dtype=out_dtype)
        y2 = lin(x, sa, out_t=buf)
        ok = ok and y2.data_ptr() == buf.data_ptr() and torch.equal(y2, y)
        try:
            lin(x, sa, out_t=torch.empty(M, N, dtype=out_dtype))
            ok = False
        except ValueError:
            pass
        # identical on every rank
        g = [torch.empty_like(y.contiguous()) for _ in range(world)]
        dist.all_gather(g, y.contiguous())
        same = all(torch.equal(g[0], t) for t in g)
        q.put((rank, bool(ok), bool(same)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("chunks,per_row,with_bias,out_dtype", [
    (1, False, False, torch.float32), (2, True, True, torch.bfloat16), (4, True, False, torch.bfloat16),
    (2, False, True, torch.float16), (4, True, True, torch.float32)])
def test_sharded_linear_world2(chunks, per_row, with_bias, out_dtype):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, chunks, per_row, with_bias, out_dtype, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0, f"rank process exited with {p.exitcode}"
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(r, True, True) for r in range(world)]


@pytest.mark.parametrize("world,chunks", [(8, 2), (8, 4), (4, 2), (4, 4)])
def test_sharded_linear_c4_partition_geometry(world, chunks):
    """BASELINE config C4's partition: N = 12288 weight rows over 8 (and 4) ranks, 2 and 4 chunk-cyclic row chunks per rank
    (nc = 768 / 384 at world 8) - the geometry the driver's multi-GPU bench runs, executed here rank for rank over gloo with
    small M and K: every rank's slab lands in its slot, gathered order equals global order, the result equals the unsharded
    fused product bit for bit (bf16, per-row scales, bias) and is identical on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    geometry = (8, 32, 12288)
    procs = [ctx.Process(target=_worker, args=(r, world, port, chunks, True, True, torch.bfloat16, q, geometry)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0, f"rank process exited with {p.exitcode}"
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(r, True, True) for r in range(world)]


def test_shard_rows_layout():
    sys.path.insert(0, PKG)
    from fp8_sharded_linear import shard_rows
    assert shard_rows(8, 2, 0).tolist() == [0, 1, 2, 3] and shard_rows(8, 2, 1).tolist() == [4, 5, 6, 7]
    assert shard_rows(8, 2, 0, chunks=2).tolist() == [0, 1, 4, 5]
    assert shard_rows(8, 2, 1, chunks=2).tolist() == [2, 3, 6, 7]
    # C4: rank 3 of 8, two chunks of 768 rows: global rows 3*768.. and 8*768 + 3*768..
    r = shard_rows(12288, 8, 3, chunks=2)
    assert r.numel() == 1536 and r[0] == 2304 and r[767] == 3071 and r[768] == 6144 + 2304 and r[-1] == 6144 + 3071
    with pytest.raises(ValueError):
        shard_rows(10, 4, 0)


def test_constructor_validates_shapes():
    """A shard whose rows do not divide into `chunks`, or scale / bias vectors of the wrong length, must raise instead
    of leaving rows of the output uninitialised."""
    sys.path.insert(0, PKG)
    from fp8_sharded_linear import ColumnShardedFP8Linear
    w = torch.zeros(30, 16, dtype=torch.uint8)
    one = torch.ones(1)
    with pytest.raises(ValueError):
        ColumnShardedFP8Linear(w, one, N=30, chunks=4)
    with pytest.raises(ValueError):
        ColumnShardedFP8Linear(w, torch.ones(7), N=30, chunks=3)
    with pytest.raises(ValueError):
        ColumnShardedFP8Linear(w, one, torch.ones(29), N=30, chunks=3)
    with pytest.raises(ValueError):
        ColumnShardedFP8Linear(w.float(), one, N=30, chunks=3)
    lin = ColumnShardedFP8Linear(w, one, torch.ones(30), N=30, chunks=3, mm=_fused_mm_transposed, out_dtype=torch.float32)
    y = lin(torch.zeros(5, 16, dtype=torch.uint8), one)      # no process group: plain single-process path
    assert y.shape == (5, 30) and torch.equal(y, torch.ones(5, 30))
