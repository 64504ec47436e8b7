"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` (backend ``nccl`` is RCCL on
ROCm; ``gloo`` for the CPU rehearsal tests).

Semantics: patches are sharded over ranks by whole baselines, parameters and
optimiser state are replicated, every loss term of the closure is a batch mean, so each rank
computes its *share* of the global loss / gradient (already divided by the global counts) and one
SUM all-reduce of the flat gradient arena and one of the loss-term vector per closure give the
global-batch result on every rank.  The upstream script has no distributed code at all.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None, device: Optional[torch.device] = None):
    """Initialise the default process group from torchrun's environment (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR/PORT).  Returns (rank, world, local_rank, group-or-None)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this driver; before HIP initialises
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # one process per GPU: make this rank's device current before anything allocates or launches
    if torch.cuda.is_available() and torch.cuda.device_count() > 0 and (backend is None or backend == "nccl"):
        torch.cuda.set_device(device if device is not None else local % torch.cuda.device_count())
    if world == 1:
        return rank, world, local, None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local, dist.group.WORLD


class Communicator:
    """RCCL communicator behind the C ABI (``lshm_comm_*``): collectives are enqueued on the caller's HIP
    stream by the library itself, so an engine can run them inside its closure, overlapped with the tail of
    the backward (``KHarmonicTrainer(process_group=...)`` attaches one when the group's backend is nccl).
    The 128-byte unique id travels through ``torch.distributed`` (any backend)."""

    def __init__(self, group=None, device: Optional[torch.device] = None):
        import ctypes as C
        from . import _lib as L
        self.lib = L.load()
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # Collective-consistent construction: rank 0's status travels WITH the id (a failure in
        # lshm_comm_unique_id raises on every rank instead of leaving the others inside the broadcast), and the
        # outcome of lshm_comm_init is MIN-reduced, so either every rank holds a communicator or none does.
        buf = C.create_string_buffer(128)
        rc0, msg0 = 0, ""
        if self.rank == 0:
            rc0 = self.lib.lshm_comm_unique_id(buf)
            if rc0:
                msg0 = self.lib.lshm_last_error_string().decode("utf-8", "replace")
        box = [(rc0, msg0, buf.raw)]
        if self.world > 1:
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        rc0, msg0, raw = box[0]
        if rc0:
            raise RuntimeError(f"lshm_amd: comm_unique_id failed on rank 0 (code {rc0}): {msg0}")
        self.device = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
        # ncclCommInitRank is itself a collective: a rank that cannot even ENTER it (library or symbols missing, device
        # not usable) would leave the others blocked inside it, so everything that can fail locally is agreed on first.
        # (A failure INSIDE the collective bring-up is bounded only by the process group's timeout.)
        local_ok = bool(self.lib.lshm_comm_available())
        if local_ok:
            try:
                with L.on_device(self.device):
                    torch.empty(1, device=self.device)
            except Exception:
                local_ok = False
        if not agree(local_ok, group if self.world > 1 else None, self.world):
            self.handle = None
            raise RuntimeError("lshm_amd: RCCL (or the device) is not usable on " + ("this rank" if not local_ok else "another rank"))
        h = C.c_void_p()
        with L.on_device(self.device):
            rc = self.lib.lshm_comm_init(raw, self.rank, self.world, C.byref(h))
        msg = self.lib.lshm_last_error_string().decode("utf-8", "replace") if rc else ""
        self.handle = h if rc == 0 else None
        if not agree(rc == 0, group if self.world > 1 else None, self.world):
            self.close()
            raise RuntimeError("lshm_amd: comm_init failed" + (f" (code {rc}): {msg}" if rc else " on another rank"))

    def allreduce_flat(self, buf: Optional[torch.Tensor], tail: Optional[torch.Tensor] = None):
        """In place SUM over ranks of a float32 buffer and / or a float64 tail, one fused launch."""
        from . import _lib as L
        with L.on_device(self.device):
            L.check(self.lib.lshm_comm_allreduce_flat(self.handle, L.ptr(buf), 0 if buf is None else buf.numel(),
                                                      L.ptr(tail), 0 if tail is None else tail.numel(),
                                                      L.stream(self.device)), "comm_allreduce_flat")

    def close(self):
        if getattr(self, "handle", None):
            self.lib.lshm_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def agree(ok: bool, group=None, world: Optional[int] = None) -> bool:
    """True iff `ok` on EVERY rank of the group (MIN all-reduce of a flag; CPU tensor for gloo, the current
    device's for nccl).  Used wherever ranks must take the same branch: which collective path a trainer uses,
    whether a communicator exists, whether the early gradient bucket is sent."""
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world <= 1 or not dist.is_initialized():
        return bool(ok)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item())


def allreduce_closure(grads: torch.Tensor, terms: torch.Tensor, group=None) -> None:
    """The two collectives of one closure: SUM over ranks of the flat gradient arena (which
    includes dM of the K-harmonic term) and of the loss-term vector."""
    if group is None and not dist.is_initialized():
        return
    dist.all_reduce(grads, op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(terms, op=dist.ReduceOp.SUM, group=group)


def allreduce_centroid_partials(num: torch.Tensor, den: torch.Tensor, group=None) -> torch.Tensor:
    """Offline centroid update (Zhang's recursion): SUM the numerator (K,D) and denominator (K)
    over ranks in one buffer, then M = num/den on every rank."""
    K, D = num.shape
    buf = torch.cat((num.reshape(-1), den.reshape(-1)))
    if group is not None or dist.is_initialized():
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf[:K * D].view(K, D) / buf[K * D:, None]


def shard_baselines(n_baselines: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) range of whole baselines for this rank (groups of the
    augmented loss never straddle ranks).  Requires an even split."""
    if n_baselines % world:
        raise ValueError(f"{n_baselines} baselines do not split evenly over {world} ranks")
    per = n_baselines // world
    return rank * per, (rank + 1) * per
