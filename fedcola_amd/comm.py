"""RCCL communicator of the C ABI (``fc_comm_*`` in include/fedcola_hip.h): one rank per process / GPU.

The unique id has to reach every rank by some side channel; ``Comm.from_torch_dist`` uses the already-initialised
``torch.distributed`` group for that (any backend -- the id is 128 bytes), ``Comm.from_file`` a shared file.  A non-PyTorch caller
does the same with its own transport.  Reference: the single-process gather of client state_dicts at
/root/reference/src/server/fedavgserver.py:566-589 and the aggregation call at :812-819."""
from __future__ import annotations

import ctypes as C
import os
import time

from . import _lib
from ._lib import check

ID_BYTES = 128


class Comm:
    def __init__(self, handle, rank, world):
        self.h, self.rank, self.world = handle, rank, world

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(ID_BYTES)
        check(_lib.lib().fc_comm_unique_id(buf, ID_BYTES))
        return buf.raw

    @classmethod
    def create(cls, uid: bytes, rank: int, world: int) -> "Comm":
        h = C.c_void_p()
        buf = C.create_string_buffer(uid, ID_BYTES)
        check(_lib.lib().fc_comm_create(buf, ID_BYTES, rank, world, C.byref(h)))
        return cls(h, rank, world)

    @classmethod
    def from_torch_dist(cls) -> "Comm":
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [None]
        if rank == 0:
            try:
                box[0] = cls.unique_id()
            except Exception as e:       # the other ranks are waiting in the broadcast: tell them instead of leaving them there
                box[0] = ("error", f"{type(e).__name__}: {e}")
        dist.broadcast_object_list(box, src=0)
        if isinstance(box[0], tuple):
            raise RuntimeError(f"rank 0 could not create an RCCL id: {box[0][1]}")
        return cls.create(box[0], rank, world)

    @classmethod
    def from_file(cls, path: str, rank: int, world: int, timeout: float = 120.0, nonce: str = None) -> "Comm":
        """Rendezvous through a shared file.  The file starts with a job nonce (``nonce``, default $FC_JOB_NONCE or
        MASTER_ADDR:MASTER_PORT of the launcher) so that a file left over by an earlier or crashed job is never taken for this job's
        id; rank 0 removes any old file first and deletes its own once every rank has joined (ncclCommInitRank is collective)."""
        import hashlib
        if nonce is None:
            nonce = os.environ.get("FC_JOB_NONCE") or f'{os.environ.get("MASTER_ADDR", "")}:{os.environ.get("MASTER_PORT", "")}'
            if nonce == ":":            # nothing identifies this job: a stale file of a crashed one would be accepted as ours
                raise ValueError("Comm.from_file needs a job nonce: pass nonce=, or set FC_JOB_NONCE or MASTER_ADDR / MASTER_PORT")
        tag = hashlib.sha256(nonce.encode()).digest()[:16]
        if rank == 0:
            try:
                os.unlink(path)
            except FileNotFoundError:
                pass
            tmp = path + ".tmp"
            with open(tmp, "wb") as f:
                f.write(tag + cls.unique_id())
            os.replace(tmp, path)
        t0 = time.time()
        uid = None
        while uid is None:
            try:
                with open(path, "rb") as f:
                    blob = f.read()
                if len(blob) == 16 + ID_BYTES and blob[:16] == tag:
                    uid = blob[16:]
                    break
            except FileNotFoundError:
                pass
            if time.time() - t0 > timeout:
                raise TimeoutError(f"no RCCL id of this job at {path}")
            time.sleep(0.01)
        comm = cls.create(uid, rank, world)
        if rank == 0:
            try:
                os.unlink(path)
            except FileNotFoundError:
                pass
        return comm

    def all_reduce(self, t):
        import torch
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        check(_lib.lib().fc_allreduce_sum(self.h, _lib.ptr(t), t.numel(), _lib.stream_ptr()))

    def close(self):
        if self.h:
            _lib.lib().fc_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
