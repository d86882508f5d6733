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
    of C^T (N, M): the GEMM writes it STRAIGHT into its slot of the gather
    buffer and the all-gather runs in place; C is returned as the `.t()` view -
    no staging tensor, no post-gather transpose or copy.
  * The epilogue stays fused and bit-identical to the single-GPU call: the
    kernels' transposed-epilogue mode (FP8MI_EPILOGUE_TRANSPOSED) takes the
    bias per output ROW (= weight row) and applies the scales in the order of
    the untransposed product, and every tile kernel adds the K-steps of an
    element in the same order - so a sharded linear returns the bits the
    unsharded fused `_scaled_mm` returns WHEN BOTH CALLS RUN UNSPLIT TILE
    KERNELS (tested, bf16 output included).  The sharded call always does
    (`split_k=1`); an unsharded AUTO call with few tokens (M = 1, the few-rows
    and skinny kernels, or a split-K launch) sums K in another order and then
    agrees only to the oracle bound, not bit for bit.
  * xGMI is point-to-point (7 links per GPU); the all-gather (each rank's slab
    to all 7 peers at once) dominates this shape, so it is pipelined against
    the GEMM: the weight rows are dealt out CHUNK-CYCLICALLY - chunk j of every
    rank forms one contiguous block of C^T - and chunk j is gathered on a side
    stream while chunk j+1 is being multiplied.  The row permutation this
    implies is applied once, at weight-load time (`shard_rows`), and it is
    chosen so that gathered order == global row order: the result needs no
    un-permutation.
  * Nothing is created per call besides the output - and not even that when
    the caller passes its own gather buffer (`forward(x, sa, out_t=...)`): the
    side stream and the per-chunk events live in the module, so a forward is
    `chunks` launches + `chunks` collectives and can be captured into a HIP graph.

  * The gather itself has two forms.  `gather="rccl"` (default): `dist.all_gather_into_tensor`, in place.  `gather="peer"`:
    the direct all-gather of include/fp8mi_peer.h - every rank stores its slab into all peers' buffers at once (all 7
    xGMI links busy, no ring), two kernel launches on the side stream and no RCCL call; the gather buffer then belongs
    to the module (IPC-mapped on every peer) and the result is a view of it, valid until the next forward.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).  The
local product is the HIP kernel; `mm` can be injected so that the sharding /
gather logic is testable with gloo on CPU (tests only - the product default
has no CPU path).  No 2/4/8-GPU measurement exists yet.
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


def _default_mm(w_rows, x_u8, scale_w, scale_x, bias, out):
    """out (rows, M) <- the transposed product W[rows] . X^T with the fused epilogue of the untransposed call."""
    import fp8_mi355x_native as native
    return native.fp8_scaled_mm(w_rows, x_u8, scale_w, scale_x, bias=bias, out_dtype=out.dtype, out=out,
                                transposed_epilogue=True, split_k=1)


class ColumnShardedFP8Linear:
    """y = x @ W^T (* scales, + bias) with W (N,K) row-sharded over the group.

    weight_u8 : this rank's shard, (N/world, K) uint8 e4m3fn bytes, rows in
                `shard_rows(N, world, rank, chunks)` order
    scale_b   : [1] or [N/world] (per local row, same order)
    bias      : None or [N/world] (same order)
    mm        : tests only - `mm(w_rows, x_u8, scale_w, scale_x, bias_or_None, out)` filling `out` (rows, M)
    gather    : "rccl" (default) or "peer" (module docstring); max_tokens sizes the module's own peer buffer;
    peer      : an existing fp8_peer_gather.PeerGather to gather through (implies gather="peer"): layers whose outputs are
                consumed one after the other can share ONE mapped buffer - the protocol orders a rank's next slab behind its
                peers' consumers of the previous one
    """

    def __init__(self, weight_u8, scale_b, bias=None, *, N: int, group=None, chunks: int = 1,
                 out_dtype=torch.bfloat16, mm=None, gather: str = "rccl", max_tokens: int | None = None, peer=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if chunks < 1 or N % (self.world * chunks):
            raise ValueError(f"N={N} must be divisible by world*chunks={self.world * chunks}")
        self.N, self.chunks, self.out_dtype = N, chunks, out_dtype
        self.nl = N // self.world
        self.nc = self.nl // chunks
        if weight_u8.dim() != 2 or weight_u8.shape[0] != self.nl or weight_u8.dtype != torch.uint8:
            raise ValueError(f"weight shard must be ({self.nl}, K) uint8")
        self.w = weight_u8
        self.scale_b = scale_b.reshape(-1)
        if self.scale_b.numel() not in (1, self.nl):
            raise ValueError(f"scale_b has {self.scale_b.numel()} elements; expected 1 or {self.nl}")
        self.bias = None if bias is None else bias.reshape(-1)
        if self.bias is not None and self.bias.numel() != self.nl:
            raise ValueError(f"bias has {self.bias.numel()} elements; expected {self.nl}")
        self.mm = mm or _default_mm
        # per-chunk views, made once
        self._w = [self.w[j * self.nc:(j + 1) * self.nc] for j in range(chunks)]
        self._sb = [self.scale_b[j * self.nc:(j + 1) * self.nc] if self.scale_b.numel() == self.nl else self.scale_b
                    for j in range(chunks)]
        self._bias = [None if self.bias is None else self.bias[j * self.nc:(j + 1) * self.nc] for j in range(chunks)]
        self._comm_stream = None
        self._events = None
        if peer is not None:
            gather = "peer"
        if gather not in ("rccl", "peer"):
            raise ValueError(f"gather must be 'rccl' or 'peer', not {gather!r}")
        if gather == "peer" and self.world < 2:
            raise ValueError("gather='peer' needs a process group of at least 2 ranks")
        self.gather, self.max_tokens, self._peer, self._own_peer = gather, max_tokens, peer, peer is None

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

    def _peer_buffer(self, M: int, dev) -> torch.Tensor:
        """gather='peer': the module's IPC-mapped gather buffer as the contiguous (N, M) C^T.  Created at the first forward
        (a collective: every rank gets here together) for max(M, max_tokens) tokens."""
        import fp8_peer_gather
        esz = torch.empty(0, dtype=self.out_dtype).element_size()
        if self._peer is None:
            self._peer = fp8_peer_gather.PeerGather(self.N * max(M, self.max_tokens or 0) * esz, dev, group=self.group)
        if self.N * M * esz > self._peer.nbytes:
            raise ValueError(f"gather='peer': {M} tokens exceed the buffer made for {self._peer.nbytes // (self.N * esz)} (pass max_tokens)")
        if (self.nc * M * esz) % 16:
            raise ValueError(f"gather='peer': a slab of {self.nc} x {M} {self.out_dtype} elements is not a multiple of 16 bytes")
        return self._peer.tensor(self.out_dtype)[:self.N * M].view(self.N, M)

    def close(self):
        """gather='peer': unmap and free the gather buffer (collective).  Nothing to do for 'rccl'."""
        if self._peer is not None and self._own_peer:
            self._peer.close()
        self._peer = None

    def forward(self, x_u8: torch.Tensor, scale_a: torch.Tensor, out_t: torch.Tensor | None = None) -> torch.Tensor:
        """x_u8 (M,K) uint8 replicated on every rank -> (M,N) `out_dtype`
        (a transposed view of the gathered (N,M) buffer), identical on every rank.
        out_t: optional caller-owned gather buffer, contiguous (N, M) of `out_dtype` on x's device (C^T): with it a
        forward allocates nothing (a forward captured into a HIP graph then owns no memory of the graph's pool, and a
        serving loop can ping-pong two buffers); the returned tensor is its `.t()` view."""
        M = x_u8.shape[0]
        dev = x_u8.device
        peer = self.gather == "peer"
        if peer:
            if out_t is not None:
                raise ValueError("gather='peer' owns its gather buffer: do not pass out_t")
            if dev.type != "cuda":
                raise ValueError("gather='peer' needs the operands on a HIP device")
            out_t = self._peer_buffer(M, dev)
        elif out_t is None:
            out_t = torch.empty(self.N, M, dtype=self.out_dtype, device=dev)  # C^T
        elif not (out_t.shape == (self.N, M) and out_t.dtype == self.out_dtype and out_t.device == dev and out_t.is_contiguous()):
            raise ValueError(f"out_t must be a contiguous ({self.N}, {M}) {self.out_dtype} tensor on {dev}")
        grouped = dist.is_initialized()          # a 1-rank group still takes the collective path
        on_gpu = dev.type == "cuda" and grouped
        if on_gpu and self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=dev)
            self._events = [torch.cuda.Event() for _ in range(self.chunks)]
        cur = torch.cuda.current_stream(dev) if on_gpu else None
        span = self.world * self.nc
        handles = []
        for j in range(self.chunks):
            block = out_t[j * span:(j + 1) * span]                     # chunk j of every rank, in rank order
            slot = block[self.rank * self.nc:(self.rank + 1) * self.nc]  # ... this rank's part: the GEMM's destination
            self.mm(self._w[j], x_u8, self._sb[j], scale_a, self._bias[j], slot)
            if not grouped:
                continue                                               # single process: the slot IS the result
            if on_gpu:
                self._events[j].record(cur)
                self._comm_stream.wait_event(self._events[j])
                if peer:                                               # slab -> the same bytes of every peer's buffer
                    self._peer.allgather(slot.data_ptr() - out_t.data_ptr(), slot.numel() * slot.element_size(), self._comm_stream.cuda_stream)
                    continue
                with torch.cuda.stream(self._comm_stream):
                    dist.all_gather_into_tensor(block, slot, group=self.group)   # in place: slot is block[rank]
            else:
                handles.append(dist.all_gather_into_tensor(block, slot, group=self.group, async_op=True))
        for h in handles:
            h.wait()
        if on_gpu:
            cur.wait_stream(self._comm_stream)
        return out_t.t()

    __call__ = forward
