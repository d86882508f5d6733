"""
Host side of include/fp8mi_peer.h: the direct (peer-store) all-gather over xGMI.

One `PeerGather` per gather buffer.  Every rank (one process per GPU) allocates a
data buffer of the same size and a flag block through libfp8mi_peer.so, the HIP IPC
handles travel through `torch.distributed`'s object all-gather (any backend: this is
setup, 128 bytes per rank), every rank maps its peers' allocations, and from then on
`allgather(offset, nbytes)` is two kernel launches on the caller's stream with no
host involvement and no RCCL call (include/fp8mi_peer.h describes the protocol and
its bounded waits).  The reference has nothing to compare with (one GPU, one command
queue: fp8_bridge.cpp:67); inside this build it is the alternative to
`dist.all_gather_into_tensor` in fp8_sharded_linear.py (`gather="peer"`).

torch is plumbing here: the process group for the handle exchange, the stream, and a
tensor VIEW of the library's allocation (`tensor()`), so that the GEMM can write its
slab straight into the gather buffer.
"""

from __future__ import annotations

import ctypes
import os

import torch
import torch.distributed as dist

HANDLE_BYTES = 64
MAX_RANKS = 16
TIMEOUT_READY = 0x1
TIMEOUT_DONE = 0x2

_lib = None


class PeerGatherError(RuntimeError):
    pass


def load():
    """libfp8mi_peer.so, next to this file.  No fallback: a missing library is an ImportError."""
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfp8mi_peer.so")
        if not os.path.exists(path):
            raise ImportError(f"{path} not found: build it with `make -C {os.path.dirname(path)}` (hipcc, gfx950)")
        lib = ctypes.CDLL(path)
        vp, i64, c_int = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        lib.fp8mi_peer_version.restype = c_int
        lib.fp8mi_peer_last_error.restype = ctypes.c_char_p
        for name, args in (("fp8mi_peer_alloc", [i64, c_int, ctypes.POINTER(vp)]), ("fp8mi_peer_free", [vp]),
                           ("fp8mi_peer_export", [vp, ctypes.c_char_p]), ("fp8mi_peer_open", [ctypes.c_char_p, ctypes.POINTER(vp)]),
                           ("fp8mi_peer_close", [vp]),
                           ("fp8mi_peer_ctx_create", [c_int, c_int, ctypes.POINTER(vp), ctypes.POINTER(vp), i64, ctypes.POINTER(vp)]),
                           ("fp8mi_peer_ctx_destroy", [vp]), ("fp8mi_peer_allgather", [vp, i64, i64, i64, vp]),
                           ("fp8mi_peer_status", [vp, vp, ctypes.POINTER(ctypes.c_uint32)])):
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = args, c_int
        _lib = lib
    return _lib


def _check(rc: int, what: str):
    if rc:
        raise PeerGatherError(f"{what} failed ({rc}): {load().fp8mi_peer_last_error().decode(errors='replace')}")


class _DeviceBytes:
    """A device allocation of the library, presented to torch through the CUDA array interface (zero copy)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class PeerGather:
    """nbytes: size of the gather buffer (the same on every rank, a multiple of 16); device: this rank's HIP device;
    group: the process group whose ranks gather (None = world).  Collective: every rank of the group constructs it at the
    same point of its program."""

    def __init__(self, nbytes: int, device: torch.device, group=None, timeout_us: int = 0):
        if not dist.is_initialized():
            raise PeerGatherError("PeerGather needs an initialised torch.distributed process group")
        self.group, self.device, self.nbytes, self.timeout_us = group, torch.device(device), int(nbytes), int(timeout_us)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        if not 2 <= self.world <= MAX_RANKS:
            raise PeerGatherError(f"PeerGather: {self.world} ranks (2..{MAX_RANKS})")
        if self.device.type != "cuda" or self.nbytes <= 0 or self.nbytes % 16:
            raise PeerGatherError(f"PeerGather: needs a HIP device and a positive multiple of 16 bytes (got {self.device}, {self.nbytes})")
        lib = load()
        self._lib, self._ctx, self._opened = lib, None, []
        self._data = self._flags = None
        with torch.cuda.device(self.device):
            data, flags = ctypes.c_void_p(), ctypes.c_void_p()
            _check(lib.fp8mi_peer_alloc(self.nbytes, 0, ctypes.byref(data)), "fp8mi_peer_alloc(data)")
            self._data = data.value
            _check(lib.fp8mi_peer_alloc(0, 1, ctypes.byref(flags)), "fp8mi_peer_alloc(flags)")
            self._flags = flags.value
            hd, hf = ctypes.create_string_buffer(HANDLE_BYTES), ctypes.create_string_buffer(HANDLE_BYTES)
            _check(lib.fp8mi_peer_export(self._data, hd), "fp8mi_peer_export(data)")
            _check(lib.fp8mi_peer_export(self._flags, hf), "fp8mi_peer_export(flags)")
            everyone = [None] * self.world
            dist.all_gather_object(everyone, (self.rank, self.nbytes, hd.raw, hf.raw), group=group)
            dptr, fptr = (ctypes.c_void_p * self.world)(), (ctypes.c_void_p * self.world)()
            for r, (their_rank, their_bytes, their_hd, their_hf) in enumerate(everyone):
                if their_rank != r or their_bytes != self.nbytes:
                    raise PeerGatherError(f"PeerGather: rank {r} announced rank {their_rank} with {their_bytes} bytes; expected {self.nbytes}")
                if r == self.rank:
                    dptr[r], fptr[r] = self._data, self._flags
                    continue
                for handle, table in ((their_hd, dptr), (their_hf, fptr)):
                    p = ctypes.c_void_p()
                    _check(lib.fp8mi_peer_open(handle, ctypes.byref(p)), f"fp8mi_peer_open(rank {r})")
                    self._opened.append(p.value)
                    table[r] = p.value
            ctx = ctypes.c_void_p()
            _check(lib.fp8mi_peer_ctx_create(self.world, self.rank, dptr, fptr, self.nbytes, ctypes.byref(ctx)), "fp8mi_peer_ctx_create")
            self._ctx = ctx.value
            self._bytes = torch.as_tensor(_DeviceBytes(self._data, self.nbytes), device=self.device)
            dist.barrier(group=group)      # nobody pushes before everybody has mapped everybody

    def tensor(self, dtype: torch.dtype = torch.uint8) -> torch.Tensor:
        """This rank's gather buffer as a flat tensor of `dtype` (a view of the library's allocation: valid until close())."""
        return self._bytes.view(dtype)

    def allgather(self, offset: int, nbytes: int, stream: int | None = None):
        """Enqueue: this rank's slab [offset, offset+nbytes) goes to every peer; behind the call on the stream every
        peer's slab of the same call is here.  `stream`: a hipStream_t as int (None = torch's current stream)."""
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        _check(self._lib.fp8mi_peer_allgather(self._ctx, int(offset), int(nbytes), self.timeout_us, stream), "fp8mi_peer_allgather")

    def status(self, stream: int | None = None) -> int:
        """Blocking: synchronise the stream, return (and clear) the TIMEOUT_* bits - 0 when every wait so far was met."""
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        word = ctypes.c_uint32()
        with torch.cuda.device(self.device):
            _check(self._lib.fp8mi_peer_status(self._ctx, stream, ctypes.byref(word)), "fp8mi_peer_status")
        return word.value

    def close(self):
        """Collective: drain, unmap the peers, free.  The tensors handed out by tensor() die here."""
        if self._ctx is None:
            return
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            dist.barrier(group=self.group)          # every rank has stopped storing into the others
            for p in self._opened:
                self._lib.fp8mi_peer_close(p)
            self._opened = []
            dist.barrier(group=self.group)          # ... and unmapped them
            self._lib.fp8mi_peer_ctx_destroy(self._ctx)
            self._ctx = None
            self._bytes = None
            self._lib.fp8mi_peer_free(self._data)
            self._lib.fp8mi_peer_free(self._flags)
            self._data = self._flags = None
