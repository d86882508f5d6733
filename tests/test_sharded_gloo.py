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


def _oracle_mm(A, B_nk, sa, sb, out_dtype, out=None):
    import fp8_oracle as o
    r = o.scaled_mm(A.numpy(), B_nk.numpy(), sa.numpy().reshape(-1), sb.numpy().reshape(-1))
    return torch.from_numpy(r).to(out_dtype)


def _worker(rank, world, port, chunks, per_row, with_bias, q):
    for p in (PKG, ORACLE):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fp8_oracle as o
        from fp8_sharded_linear import ColumnShardedFP8Linear, shard_rows
        rng = np.random.default_rng(42)  # same data on every rank
        M, K, N = 24, 64, 32 * world
        x = torch.from_numpy(rng.integers(0, 256, size=(M, K), dtype=np.uint8))
        W = torch.from_numpy(rng.integers(0, 256, size=(N, K), dtype=np.uint8))
        sa = torch.tensor([0.03])
        sb = torch.from_numpy(rng.uniform(0.01, 0.05, size=N).astype(np.float32)) if per_row else torch.tensor([0.02])
        bias = torch.from_numpy(rng.standard_normal(N).astype(np.float32)) if with_bias else None
        lin = ColumnShardedFP8Linear.from_full(W, sb, bias, chunks=chunks, out_dtype=torch.float32, mm=_oracle_mm)
        # every rank owns N/world rows, all ranks together cover each row exactly once
        rows = [shard_rows(N, world, r, chunks) for r in range(world)]
        assert sorted(torch.cat(rows).tolist()) == list(range(N))
        y = lin(x, sa)
        assert y.shape == (M, N) and y.stride() == (1, M)  # .t() view of the gathered C^T
        ref = o.scaled_mm(x.numpy(), W.numpy(), sa.numpy(), sb.numpy(), accumulate="f64")
        bound = o.abs_dot_bound(x.numpy(), W.numpy(), sa.numpy(), sb.numpy())
        if bias is not None:
            ref = ref + bias.numpy()[None, :]
            bound = bound + np.abs(bias.numpy())[None, :]
        # float32 products summed in another order (transposed blocks): fp32 rounding only
        ok = bool(np.all(np.abs(y.numpy() - ref) <= 4e-6 * bound + 1e-30))
        # identical on every rank
        g = [torch.empty_like(y.contiguous()) for _ in range(world)]
        dist.all_gather(g, y.contiguous())
        same = all(torch.equal(g[0], t) for t in g)
        q.put((rank, bool(ok), bool(same)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("chunks,per_row,with_bias", [(1, False, False), (2, True, True), (4, True, False)])
def test_sharded_linear_world2(chunks, per_row, with_bias):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, chunks, per_row, with_bias, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0, f"rank process exited with {p.exitcode}"
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(r, True, True) for r in range(world)]


def test_shard_rows_layout():
    sys.path.insert(0, PKG)
    from fp8_sharded_linear import shard_rows
    assert shard_rows(8, 2, 0).tolist() == [0, 1, 2, 3] and shard_rows(8, 2, 1).tolist() == [4, 5, 6, 7]
    assert shard_rows(8, 2, 0, chunks=2).tolist() == [0, 1, 4, 5]
    assert shard_rows(8, 2, 1, chunks=2).tolist() == [2, 3, 6, 7]
    with pytest.raises(ValueError):
        shard_rows(10, 4, 0)
