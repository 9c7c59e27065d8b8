"""tt_comm_* of the C ABI (csrc/comm.cpp: RCCL bound at run time) behind a small Python class.

The reference has no communication of any kind; this is the transport a non-torch
binder of include/tt_hotpath.h would use for the row-sharded step, and the module path (parallel.py /
collectives.py) can run on it instead of `torch.distributed`'s process group:
`collectives.use_native_transport(NativeComm.from_torch_distributed(device))` (bench.py `--transport native`);
`use_native_transport(None)` closes it.  One communicator per process = per GPU.  The 128-byte RCCL id is created by rank 0
(`NativeComm.unique_id()`) and handed to the other ranks over ANY host channel; `from_torch_distributed`
uses the already initialised process group (gloo or nccl) for exactly that one broadcast.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _native as N

_DTYPES = {torch.float32: N.TT_COMM_F32, torch.int32: N.TT_COMM_I32, torch.int64: N.TT_COMM_I64, torch.uint8: N.TT_COMM_U8}


def _dt(t: torch.Tensor) -> int:
    try:
        return _DTYPES[t.dtype]
    except KeyError:
        raise TypeError(f"tt_comm: unsupported dtype {t.dtype} (float32, int32, int64, uint8)") from None


class NativeComm:
    def __init__(self, id_bytes: bytes, rank: int, world: int, device: torch.device):
        if len(id_bytes) != N.TT_COMM_ID_BYTES:
            raise ValueError(f"the communicator id is {N.TT_COMM_ID_BYTES} bytes")
        if device.type != "cuda":
            raise RuntimeError("NativeComm needs an MI355X device; there is no CPU path")
        self.lib = N.load()
        self.rank, self.world, self.device = rank, world, device
        handle = C.c_void_p()
        buf = C.create_string_buffer(id_bytes, N.TT_COMM_ID_BYTES)
        with torch.cuda.device(device):  # RCCL binds the communicator to the current device
            N.check(self.lib.tt_comm_init(buf, rank, world, C.byref(handle)), "tt_comm_init")
        self._h: Optional[C.c_void_p] = handle

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(N.TT_COMM_ID_BYTES)
        N.check(N.load().tt_comm_unique_id(buf), "tt_comm_unique_id")
        return buf.raw

    @classmethod
    def from_torch_distributed(cls, device: torch.device) -> "NativeComm":
        """Create the communicator of every rank of the initialised torch.distributed group; the group is used
        for ONE broadcast of the id (host memory under gloo, device memory under nccl)."""
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        on_dev = dist.get_backend() == "nccl"
        t = torch.zeros(N.TT_COMM_ID_BYTES, dtype=torch.uint8, device=device if on_dev else "cpu")
        if rank == 0:
            t.copy_(torch.frombuffer(bytearray(cls.unique_id()), dtype=torch.uint8))
        if world > 1:
            dist.broadcast(t, src=0)
        return cls(bytes(t.cpu().tolist()), rank, world, device)

    def close(self) -> None:
        """Destroy the RCCL communicator.  Called by collectives.use_native_transport(None) / at interpreter exit (registered
        by `collectives.use_native_transport`), i.e. BEFORE torch.distributed's process group goes away."""
        if self._h is not None:
            h, self._h = self._h, None
            N.check(self.lib.tt_comm_destroy(h), "tt_comm_destroy")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        """(rank, world) -- world as RCCL itself reports it."""
        r, w = C.c_int32(), C.c_int32()
        N.check(self.lib.tt_comm_size(self._h, C.byref(r), C.byref(w)), "tt_comm_size")
        return r.value, w.value

    # every call is asynchronous on `stream` (default: torch's current stream)
    @staticmethod
    def _s(stream) -> int:
        return (stream or torch.cuda.current_stream()).cuda_stream

    def all_to_all(self, send: torch.Tensor, recv: Optional[torch.Tensor] = None, stream=None) -> torch.Tensor:
        """Chunk r of `send` (equal chunks along dim 0) goes to rank r; chunk r of the result came from rank r."""
        send = send.contiguous()
        if send.shape[0] % self.world:
            raise ValueError("all_to_all: dim 0 must be a multiple of the world size")
        recv = torch.empty_like(send) if recv is None else recv
        N.check(self.lib.tt_comm_alltoall(self._h, send.data_ptr(), recv.data_ptr(), send.numel() // self.world, _dt(send),
                                          self._s(stream)), "tt_comm_alltoall")
        return recv

    def all_gather(self, send: torch.Tensor, recv: Optional[torch.Tensor] = None, stream=None) -> torch.Tensor:
        send = send.contiguous()
        if recv is None:
            recv = send.new_empty((self.world * send.shape[0],) + tuple(send.shape[1:]))
        N.check(self.lib.tt_comm_allgather(self._h, send.data_ptr(), recv.data_ptr(), send.numel(), _dt(send),
                                           self._s(stream)), "tt_comm_allgather")
        return recv

    def reduce_scatter(self, send: torch.Tensor, recv: Optional[torch.Tensor] = None, op: int = N.TT_COMM_SUM,
                       stream=None) -> torch.Tensor:
        send = send.contiguous()
        if recv is None:
            recv = send.new_empty((send.shape[0] // self.world,) + tuple(send.shape[1:]))
        N.check(self.lib.tt_comm_reduce_scatter(self._h, send.data_ptr(), recv.data_ptr(), recv.numel(), _dt(send), op,
                                                self._s(stream)), "tt_comm_reduce_scatter")
        return recv

    def all_reduce_(self, x: torch.Tensor, op: int = N.TT_COMM_SUM, stream=None) -> torch.Tensor:
        if not x.is_contiguous():
            raise ValueError("all_reduce_: contiguous tensor expected (in-place)")
        N.check(self.lib.tt_comm_allreduce(self._h, x.data_ptr(), x.data_ptr(), x.numel(), _dt(x), op, self._s(stream)),
                "tt_comm_allreduce")
        return x

    def broadcast_(self, x: torch.Tensor, root: int = 0, stream=None) -> torch.Tensor:
        if not x.is_contiguous():
            raise ValueError("broadcast_: contiguous tensor expected (in-place)")
        N.check(self.lib.tt_comm_broadcast(self._h, x.data_ptr(), x.numel(), _dt(x), root, self._s(stream)),
                "tt_comm_broadcast")
        return x
