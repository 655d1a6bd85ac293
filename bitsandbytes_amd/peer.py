"""One-shot all-gather over peer-mapped buffers — the latency-bound exchange step of the N-sharded 4-bit linear layer.

``ShardedLinear4bit`` (``parallel.py``) reassembles ``y`` with one all-gather per layer. At decode sizes that message is
2.7 KB per rank (M = 1, Llama FFN shard): a ring collective pays its whole protocol for nothing. :class:`PeerAllGather`
instead maps every rank's gather buffer into every process once (hipIpc, exchanged through the ordinary process group) and
then runs ONE kernel per collective in which each rank stores its shard straight into its slot of every peer's buffer over
xGMI, publishes a flag and waits for the peers' flags (``csrc/peer_gather.hip``). It is a plain stream-ordered launch: no host
synchronisation, capturable in a hipGraph with the kernels around it (``parallel.GraphedBlock``).

A wait that runs into its bound does not hang the queue: the launch ends, the missing rank's rows of the result are NaN (all-ones
bytes) and a sticky status word is set. Nothing on the device raises: callers poll :meth:`PeerAllGather.check` at their natural
synchronisation points (``parallel.GraphedBlock`` does every ``check_every`` replays; ``bench.py`` before it trusts the kernel).

Nothing in the reference to mirror (it has no collective code). Exercised here with two processes sharing one GPU
and in-process at world size 1; **not measured on a multi-GPU node by us**.
"""
from __future__ import annotations

import ctypes as ct
from typing import Optional

import torch
import torch.distributed as dist

from .cextension import lib

MAX_WORLD = 8
DEFAULT_MAX_BYTES = 64 * 1024  # per rank; larger messages are bandwidth-bound: use the group's own all-gather


class PeerAllGather:
    """``all_gather(y_local) -> [G * m, ns]`` (rank-major, the layout of ``dist.all_gather_into_tensor``) for shards of at
    most ``max_bytes`` bytes. Every rank of ``group`` constructs one (collectively: the handles travel through the group) and
    calls ``all_gather`` the same number of times with the same shape."""

    def __init__(self, group=None, max_bytes: int = DEFAULT_MAX_BYTES, device: Optional[torch.device] = None):
        if not dist.is_initialized():
            raise RuntimeError("PeerAllGather needs an initialised process group (the buffer handles travel through it)")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if self.world > MAX_WORLD:
            raise ValueError(f"PeerAllGather serves the GPUs of one node (<= {MAX_WORLD} ranks), got {self.world}")
        self.max_bytes = int(max_bytes)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._mapped = []
        self._local = None
        # Every step below is collective and none may leave the ranks disagreeing about whether the object exists: failures are
        # carried to the exchanges as values, and every rank raises - or none does.
        handle = ct.create_string_buffer(64)
        with torch.cuda.device(self.device):
            nbytes = lib.bnb_mi355x_peer_buffer_bytes(self.world, self.max_bytes)
            self._local = lib.bnb_mi355x_peer_alloc(nbytes)
            exported = bool(self._local) and lib.bnb_mi355x_peer_export(ct.c_void_p(self._local), handle) == 0
        everyone = [None] * self.world
        dist.all_gather_object(everyone, handle.raw if exported else None, group=group)
        ptrs, problem = [], None
        if any(raw is None for raw in everyone):
            problem = "a rank could not allocate or export its fine-grained buffer"
        else:
            with torch.cuda.device(self.device):
                for r, raw in enumerate(everyone):
                    if r == self.rank:
                        ptrs.append(self._local)
                        continue
                    p = lib.bnb_mi355x_peer_open(ct.create_string_buffer(raw, 64))
                    if not p:
                        problem = f"rank {self.rank} could not map the buffer of rank {r} (no peer access between the devices?)"
                        break
                    self._mapped.append(p)
                    ptrs.append(p)
        problems = [None] * self.world
        dist.all_gather_object(problems, problem, group=group)  # also: nobody stores into a buffer its owner has not zeroed yet
        problems = [q for q in problems if q]
        if problems:
            self._release()
            raise RuntimeError("PeerAllGather: " + "; ".join(problems))
        self._bufs = (ct.c_void_p * self.world)(*ptrs)

    def _release(self) -> None:
        for p in self._mapped:
            lib.bnb_mi355x_peer_close(ct.c_void_p(p))
        if self._local:
            lib.bnb_mi355x_peer_free(ct.c_void_p(self._local))
        self._mapped, self._local = [], None

    def all_gather(self, y_local: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        y2 = y_local.reshape(-1, y_local.shape[-1]).contiguous()
        nbytes = y2.numel() * y2.element_size()
        if nbytes > self.max_bytes:
            raise ValueError(f"shard of {nbytes} bytes exceeds this PeerAllGather's max_bytes ({self.max_bytes})")
        if y2.device != self.device:
            raise ValueError(f"tensor on {y2.device}, buffers on {self.device}")
        if out is None:
            out = torch.empty((self.world * y2.shape[0], y2.shape[1]), dtype=y2.dtype, device=y2.device)
        elif out.numel() * out.element_size() != self.world * nbytes or not out.is_contiguous():
            raise ValueError("out must be a contiguous tensor of world x shard elements")
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            lib.bnb_mi355x_peer_allgather(self._bufs, self.world, self.rank, ct.c_void_p(y2.data_ptr()), ct.c_void_p(out.data_ptr()),
                                          nbytes, self.max_bytes, ct.c_void_p(stream))
        return out

    def status(self) -> int:
        """0 while every wait of every collective found its peer, 1 once one gave up (synchronises the device)."""
        return int(lib.bnb_mi355x_peer_status(ct.c_void_p(self._local)))

    def check(self) -> None:
        """Raises if any collective so far gave up waiting for a peer (synchronises the device)."""
        st = self.status()
        if st != 0:
            raise RuntimeError("PeerAllGather: a rank did not arrive at a collective within the wait bound (status %d)" % st)

    def close(self) -> None:
        if self._local is None:
            return
        torch.cuda.synchronize(self.device)
        try:
            dist.barrier(group=self.group)  # no peer is still storing into a buffer that is about to go away
        except Exception:  # the group may already be gone at interpreter exit
            pass
        self._release()

    def __del__(self):
        # (garbage collection is not collective: no barrier here. The local work is drained before the buffer goes away; peers
        # that may still store into it are the reason `close()` exists - an object that is dropped while the process group is
        # alive and no barrier has run keeps its memory mapped rather than freeing it under a peer's stores)
        try:
            if self._local is None:
                return
            torch.cuda.synchronize(self.device)
            if self.world > 1 and dist.is_initialized():
                self._mapped, self._local = [], None   # leaked on purpose: see above
                return
            self._release()
        except Exception:
            pass
