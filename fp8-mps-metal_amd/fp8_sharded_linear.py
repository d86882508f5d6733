"""
N-column-sharded FP8 linear with an all-gather over RCCL / xGMI.

The reference has no multi-device code at all (one Apple GPU, one command
queue - fp8_bridge.cpp:67); this module is the build's only distributed piece
(BASELINE.json configs[3]).  The op it shards is exactly
`_metal_scaled_mm` (fp8_mps_patch.py:53-106): C = (A . B^T) * sa * sb (+ bias).

Why it shards this way (MI355X-first):

  * Output columns are independent, so each rank multiplies the replicated
    activations A (M,K) with ITS rows of the weight B (N,K) - no communication
    during compute; the only collective is the all-gather of the output.
  * Each rank computes the TRANSPOSED block  C^T[rows] = B[rows] . A^T  - free,
    because both operands are K-contiguous and the kernel does not care which
    one is called "A".  The per-rank result (rows, M) is then a contiguous slab
    of C^T (N, M), so `all_gather_into_tensor` lands the full C^T in place and
    C is returned as its `.t()` view: no post-gather transpose or copy.
  * xGMI is point-to-point (7 links per GPU); the all-gather (each rank's slab
    to all 7 peers at once) dominates this shape, so it is pipelined against
    the GEMM: the weight rows are dealt out CHUNK-CYCLICALLY - chunk j of every
    rank forms one contiguous block of C^T - and chunk j is gathered on a side
    stream while chunk j+1 is being multiplied.  The row permutation this
    implies is applied once, at weight-load time (`shard_rows`), and it is
    chosen so that gathered order == global row order: the result needs no
    un-permutation.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).  The
local product is the HIP kernel; `mm` can be injected so that the sharding /
gather logic is testable with gloo on CPU (tests only - the product default
has no CPU path).
"""

from __future__ import annotations

import torch
import torch.distributed as dist


def shard_rows(N: int, world: int, rank: int, chunks: int = 1) -> torch.Tensor:
    """Global weight-row (= output-column) indices owned by `rank`, in the order
    its local shard stores them.  Chunk-cyclic: with Nc = N / (world*chunks),
    local row j*Nc + i  <->  global row j*world*Nc + rank*Nc + i."""
    if N % (world * chunks):
        raise ValueError(f"N={N} must be divisible by world*chunks={world * chunks}")
    nc = N // (world * chunks)
    j = torch.arange(chunks).repeat_interleave(nc)
    i = torch.arange(nc).repeat(chunks)
    return j * (world * nc) + rank * nc + i


def _default_mm(A, B_nk, sa, sb, out_dtype, out=None):
    import fp8_mi355x_native as native
    r = native.fp8_scaled_mm(A, B_nk, sa, sb, out_dtype=out_dtype)
    if out is not None:
        out.copy_(r)
        return out
    return r


class ColumnShardedFP8Linear:
    """y = x @ W^T (* scales, + bias) with W (N,K) row-sharded over the group.

    weight_u8 : this rank's shard, (N/world, K) uint8 e4m3fn bytes, rows in
                `shard_rows(N, world, rank, chunks)` order
    scale_b   : [1] or [N/world] (per local row, same order)
    bias      : None or [N/world] (same order)
    """

    def __init__(self, weight_u8, scale_b, bias=None, *, N: int, group=None, chunks: int = 1,
                 out_dtype=torch.bfloat16, mm=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.N, self.chunks, self.out_dtype = N, chunks, out_dtype
        self.nl = N // self.world
        self.nc = self.nl // chunks
        if weight_u8.shape[0] != self.nl or weight_u8.dtype != torch.uint8:
            raise ValueError(f"weight shard must be ({self.nl}, K) uint8")
        self.w = weight_u8
        self.scale_b = scale_b.reshape(-1)
        self.bias = None if bias is None else bias.reshape(-1)
        self.mm = mm or _default_mm
        self._comm_stream = None

    @classmethod
    def from_full(cls, weight_u8_full, scale_b, bias=None, *, group=None, chunks=1, **kw):
        """Convenience for tests / weight loading: slice this rank's rows out of
        the full (N,K) weight (and per-row scale / bias)."""
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        N = weight_u8_full.shape[0]
        rows = shard_rows(N, world, rank, chunks).to(weight_u8_full.device)
        sb = scale_b.reshape(-1)
        sb = sb[rows] if sb.numel() == N else sb
        b = None if bias is None else bias.reshape(-1)[rows]
        return cls(weight_u8_full[rows].contiguous(), sb, b, N=N, group=group, chunks=chunks, **kw)

    def forward(self, x_u8: torch.Tensor, scale_a: torch.Tensor) -> torch.Tensor:
        """x_u8 (M,K) uint8 replicated on every rank -> (M,N) `out_dtype`
        (a transposed view of the gathered (N,M) buffer), identical on every rank."""
        M = x_u8.shape[0]
        dev = x_u8.device
        out_t = torch.empty(self.N, M, dtype=self.out_dtype, device=dev)  # C^T
        grouped = dist.is_initialized()          # a 1-rank group still takes the collective path
        on_gpu = dev.type == "cuda" and grouped
        if on_gpu and self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=dev)
        handles = []
        for j in range(self.chunks):
            lo, hi = j * self.nc, (j + 1) * self.nc
            sb = self.scale_b[lo:hi] if self.scale_b.numel() == self.nl else self.scale_b
            # transposed product: the weight rows play "A", the activations play "B_nk"
            part = self.mm(self.w[lo:hi], x_u8, sb, scale_a, self.out_dtype)
            if self.bias is not None:
                part.add_(self.bias[lo:hi].to(part.dtype)[:, None])
            block = out_t[j * self.world * self.nc:(j + 1) * self.world * self.nc]
            if not grouped:
                block.copy_(part)
            elif on_gpu:
                ev = torch.cuda.Event()
                ev.record()
                with torch.cuda.stream(self._comm_stream):
                    self._comm_stream.wait_event(ev)
                    dist.all_gather_into_tensor(block, part, group=self.group)
                    part.record_stream(self._comm_stream)
            else:
                handles.append(dist.all_gather_into_tensor(block, part.contiguous(), group=self.group, async_op=True))
        for h in handles:
            h.wait()
        if on_gpu:
            torch.cuda.current_stream(dev).wait_stream(self._comm_stream)
        return out_t.t()

    __call__ = forward
