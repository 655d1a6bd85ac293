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
import os
from typing import Optional

import torch
import torch.distributed as dist

from .cextension import lib

MAX_WORLD = 8
DEFAULT_MAX_BYTES = 64 * 1024  # per rank; larger messages are bandwidth-bound: use the group's own all-gather


class _PeerBuffers:
    """One fine-grained device buffer per rank, mapped into every process of the group (hipIpc; the handles travel through the
    ordinary process group). Construction is collective."""

    def __init__(self, nbytes_of, group=None, device: Optional[torch.device] = None, alloc=None):
        name = type(self).__name__
        if not dist.is_initialized():
            raise RuntimeError(f"{name} needs an initialised process group (the buffer handles travel through it)")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if self.world > MAX_WORLD:
            raise ValueError(f"{name} serves the GPUs of one node (<= {MAX_WORLD} ranks), got {self.world}")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._mapped = []
        self._local = None
        # (`alloc` may be a factory that needs the group - the chain picks its memory kind from the topology: called here, on
        # every rank, before anything is allocated)
        alloc = alloc(self) if getattr(alloc, "needs_buffers", False) else (alloc or lib.bnb_mi355x_peer_alloc)
        # Every step below is collective and none may leave the ranks disagreeing about whether the object exists: failures are
        # carried to the exchanges as values, and every rank raises - or none does.
        handle = ct.create_string_buffer(64)
        with torch.cuda.device(self.device):
            nbytes = nbytes_of(self.world)
            self._local = alloc(nbytes)
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
            raise RuntimeError(f"{name}: " + "; ".join(problems))
        self._bufs = (ct.c_void_p * self.world)(*ptrs)

    def _release(self) -> None:
        for p in self._mapped:
            lib.bnb_mi355x_peer_close(ct.c_void_p(p))
        if self._local:
            lib.bnb_mi355x_peer_free(ct.c_void_p(self._local))
        self._mapped, self._local = [], None

    def status(self) -> int:
        """0 while every wait of every launch found its peer, 1 once one gave up (synchronises the device)."""
        return int(lib.bnb_mi355x_peer_status(ct.c_void_p(self._local)))

    def check(self) -> None:
        """Raises if any launch so far gave up waiting for a peer (synchronises the device)."""
        st = self.status()
        if st != 0:
            raise RuntimeError(f"{type(self).__name__}: a rank did not arrive within the wait bound (status {st})")

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


class PeerAllGather(_PeerBuffers):
    """``all_gather(y_local) -> [G * m, ns]`` (rank-major, the layout of ``dist.all_gather_into_tensor``) for shards of at
    most ``max_bytes`` bytes. Every rank of ``group`` constructs one (collectively: the handles travel through the group) and
    calls ``all_gather`` the same number of times with the same shape."""

    def __init__(self, group=None, max_bytes: int = DEFAULT_MAX_BYTES, device: Optional[torch.device] = None):
        self.max_bytes = int(max_bytes)
        super().__init__(lambda world: lib.bnb_mi355x_peer_buffer_bytes(world, self.max_bytes), group, device)

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


def _ranks_on_my_device(group, device) -> int:
    """How many ranks of the group use the device this rank uses (1 on a real node; > 1 where processes share a GPU - the only
    way a 1-GPU box can run the peer paths). Identified by hostname + PCI address where torch reports it, else by hostname +
    device index + HIP_VISIBLE_DEVICES. Collective: a rank that cannot identify its device says so THROUGH the exchange, and every
    rank raises - none is left waiting in the all-gather for a peer that raised alone."""
    import socket

    try:
        props = torch.cuda.get_device_properties(device)
        pci = tuple(getattr(props, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
        key = (socket.gethostname(), pci) if all(v is not None for v in pci) else \
            (socket.gethostname(), torch.device(device).index, os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES"))
    except Exception as exc:  # noqa: BLE001  (carried to the exchange below)
        key = ("cannot identify the device", f"rank {dist.get_rank(group)}: {type(exc).__name__}: {exc}")
    keys = [None] * dist.get_world_size(group)
    dist.all_gather_object(keys, key, group=group)
    failed = [k[1] for k in keys if k and k[0] == "cannot identify the device"]
    if failed:
        raise RuntimeError("a rank could not identify its device: " + "; ".join(failed))
    return sum(1 for k in keys if k == key)


class PeerChain(_PeerBuffers):
    """The all-gather of an N-sharded decode layer FUSED into the gemv launches on either side of it (``csrc/gemv4_stream.hip``,
    ``PeerChain``): ``gemv(..., produce=True)`` stores this rank's ``ns`` outputs straight into every rank's exchange buffer as
    8-byte {two values, tag} granules; the next layer's ``gemv(None, ..., consume=True)`` takes its ``x`` from there - fetched
    behind its own weight requests, re-fetched until the tags are there - and ``read`` ENDS a chain with the plain tensor (every
    chain must end with it, a captured block in particular: it is the launch that advances the buffers' epoch). One
    launch per layer, no collective launch, no host synchronisation; capturable in a hipGraph. M = 1, fp16 / bf16.

    Every rank issues the same sequence of calls; a chain holds at least two exchanges (two ``produce`` launches) before its
    ``read``. Results are bit-identical to ``ShardedLinear4bit`` layer by layer (the kernel's
    arithmetic does not depend on the launch geometry). ``max_values``: the longest gathered vector (``world * ns``) and the
    longest consumed ``x`` (``K``) the chain will see. Exercised between processes sharing one GPU; **not measured on a
    multi-GPU node by us**."""

    def __init__(self, group=None, max_values: int = 32768, device: Optional[torch.device] = None, self_test: bool = True,
                 memory: Optional[str] = None):
        # (in fours: an even number of granules per region keeps every region 16-byte aligned - the two-granule stores and the
        # 16-byte fetches of the kernel need it)
        self.max_values = (int(max_values) + 3) & ~3
        if memory is None:
            memory = os.environ.get("BNB_MI355X_PEER_CHAIN_MEMORY") or None
        if memory not in (None, "fine", "coarse"):
            raise ValueError(f"memory must be 'fine', 'coarse' or None (by topology), got {memory!r}")

        def make_alloc(bufs):
            # Which memory the exchange buffers live in. Remote GPUs store into a rank's buffer WHILE a kernel of that rank polls
            # it; HIP makes coarse-grained (ordinary hipMalloc) memory coherent across devices at kernel boundaries only, so across
            # a link the buffers must be FINE-grained (what RCCL's LL protocol and PeerAllGather use) - a stale L2 line would make
            # every consumer spin to its bound and emit NaN. Ordinary cacheable memory is kept for the one topology where nothing
            # crosses a link: every rank on ONE device (processes sharing a GPU - the 1-GPU test set-up - or a group of one).
            # The decision is collective (an all-gather of device identities), so every rank allocates the same kind.
            bufs.sharing = _ranks_on_my_device(bufs.group, bufs.device)
            bufs.memory = memory or ("coarse" if bufs.sharing >= bufs.world else "fine")
            fine = 1 if bufs.memory == "fine" else 0
            return lambda nbytes: lib.bnb_mi355x_peer_chain_alloc(nbytes, fine)

        make_alloc.needs_buffers = True
        super().__init__(lambda world: lib.bnb_mi355x_peer_chain_buffer_bytes(self.max_values), group, device, alloc=make_alloc)
        # ranks that share one device must all be resident at once (a launch that waits for its peers may not fill the device
        # alone): each gets its share of the CUs. One rank per device - the real case - gets them all.
        cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        self.wg_limit = 0 if self.sharing <= 1 else max(1, cus // self.sharing)
        # exchanges produced since the last read-out: the buffer's epoch word lives on the device and only `read` advances it
        # (a hipGraph captures these offsets; replayed, they are relative to an epoch that has moved on by a whole chain)
        self._pending = 0
        self._epoch = torch.zeros(64, dtype=torch.int32, device=self.device)  # ordinary (cacheable) memory: only this rank's launches touch it
        self._broken = None
        if self_test:
            self._self_test()

    def _self_test(self) -> None:
        """Collective start-up check: two chains of three small sharded layers each (produce -> consume + produce -> consume +
        produce -> read-out) against the same layers with the group's own all-gather between them, bit for bit, on every rank.
        The chain's transport has only ever run between processes sharing one GPU by its authors; a node on which it does not
        reproduce the collective (memory kind, peer access, a stack that reorders what it may not) must not get silent NaNs or
        stale activations out of ``ShardedLinear4bitChain`` - every rank raises here, or none does."""
        from . import functional as F
        from .autograd import matmul_4bit
        from .parallel import shard_quant_state

        world, rank, dev = self.world, self.rank, self.device
        # gathered width H = world * ns: a multiple of 64 (blocksize) and of 2 * world (row pairs), within this chain's max_values
        step = 64
        while step % (2 * world):
            step += 64
        H = (min(self.max_values, 512) // step) * step
        problem = None
        ok = True
        if H == 0:
            return  # (max_values below one test layer: nothing this chain could carry would fit either)
        ns = H // world
        # EVERY host collective below runs on EVERY rank whatever happened locally - a rank that left the sequence on an exception
        # would meet its peers' all-gathers with the final vote (mismatched collectives: a hang on RCCL). Local failures only set
        # `ok` / `problem`; the device-side exchange of a rank that cannot launch is simply missing, which its peers' bounded waits
        # turn into a time-out and a "no" vote.

        def guarded(fn, fallback=None):
            nonlocal ok, problem
            try:
                return fn()
            except Exception as exc:  # noqa: BLE001
                if ok:
                    ok, problem = False, f"{type(exc).__name__}: {exc}"
                return fallback

        with torch.no_grad(), torch.cuda.device(dev):
            gen = guarded(lambda: torch.Generator(device=dev).manual_seed(20250922))  # the same weights and inputs on every rank

            def make_layer(i):
                W = (torch.randn(H, H, device=dev, generator=gen) / H**0.5).to(torch.bfloat16)
                packed, st = F.quantize_4bit(W, blocksize=64, quant_type="nf4" if i != 1 else "fp4", compress_statistics=False)
                return shard_quant_state(packed, st, rank, world)

            layers = [guarded(lambda i=i: make_layer(i)) for i in range(3)]
            # `launch`: decided from shapes alone, so identical on every rank. A rank whose RESULT is wrong keeps launching like
            # the others (its peers consume its granules: a rank that stopped would make them spin to their bound); only a
            # refusal - which every rank sees alike - stops the launches.
            launch = bool(guarded(lambda: all(self.serves(ns, H, 64, consume=i > 0) for i in range(3)), False))
            if not launch and ok:
                ok, problem = False, f"the fused form refuses the self-test shapes ({ns} x {H})"
            nccl = dist.get_backend(self.group) == "nccl"
            zeros = torch.zeros(ns, dtype=torch.bfloat16, device=dev)
            for rep in range(2):  # (the second chain re-uses the regions of the first under a new epoch)
                x = guarded(lambda: torch.randn(H, device=dev, generator=gen).to(torch.bfloat16), torch.zeros(H, dtype=torch.bfloat16, device=dev))
                want = x
                for layer in layers:
                    y_loc = guarded(lambda: matmul_4bit(want.view(1, H), layer[0], quant_state=layer[1]).reshape(-1).contiguous(), zeros) \
                        if layer is not None else zeros
                    if nccl:
                        buf = torch.empty(H, dtype=torch.bfloat16, device=dev)
                        dist.all_gather_into_tensor(buf, y_loc, group=self.group)
                    else:  # (a host-side group - gloo in the shared-GPU test set-up: through the host)
                        parts = [None] * world
                        dist.all_gather_object(parts, y_loc.cpu(), group=self.group)
                        buf = torch.cat(parts).to(dev)
                    want = buf
                if launch and ok:
                    def run_chain():
                        for i, (q, st) in enumerate(layers):
                            if not self.gemv(x if i == 0 else None, q, st, consume=i > 0, produce=True, dtype=torch.bfloat16):
                                raise RuntimeError("a launch of the self-test chain was refused")  # (cannot happen behind serves() + aligned shards)
                        got = self.read(H, torch.bfloat16)
                        torch.cuda.synchronize(dev)
                        return got

                    got = guarded(run_chain)
                    if got is not None and ok:
                        if self.status() != 0:
                            ok, problem = False, "a wait ran into its bound (a peer's granules never became visible)"
                        elif not torch.equal(got, want):
                            ok, problem = False, "its result differs from the layers with the group's all-gather between them"
        votes = [None] * world
        dist.all_gather_object(votes, None if ok else f"rank {rank}: {problem}", group=self.group)
        votes = [v for v in votes if v]
        if votes:
            self._release_after_failed_test()
            raise RuntimeError(f"PeerChain self-test failed on this node ({self.memory}-grained buffers, {self.sharing} rank(s) per device) - "
                               "use ShardedLinear4bit (a kernel + an all-gather per layer) instead: " + "; ".join(votes))

    def _release_after_failed_test(self) -> None:
        try:
            torch.cuda.synchronize(self.device)
            dist.barrier(group=self.group)
        except Exception:  # noqa: BLE001
            pass
        self._release()

    @staticmethod
    def _dt(dtype: torch.dtype) -> int:
        if dtype == torch.float16:
            return 1
        if dtype == torch.bfloat16:
            return 2
        raise ValueError(f"PeerChain serves fp16 / bf16 activations, got {dtype}")

    def serves(self, ns: int, K: int, blocksize: int, consume: bool, produce: bool = True, gated: bool = False) -> bool:
        """The shape preconditions of the fused form - the launcher's OWN check (``peer_geometry`` in csrc/gemv4_stream.hip, launch
        geometry included), asked through ``bnb_mi355x_gemv_4bit_peer_serves``: nothing here can drift from what a launch accepts.
        Shapes only (and the chain's own constants), so every rank answers the same."""
        with torch.cuda.device(self.device):  # (the geometry depends on the device's CU count)
            return bool(lib.bnb_mi355x_gemv_4bit_peer_serves(self.world, int(ns), int(K), int(blocksize),
                                                             (1 if consume else 0) | (2 if produce else 0) | (8 if gated else 0),
                                                             self.max_values, self.wg_limit))

    def gemv(self, x: Optional[torch.Tensor], packed: torch.Tensor, quant_state, bias: Optional[torch.Tensor] = None,
             out_local: Optional[torch.Tensor] = None, consume: bool = False, produce: bool = True,
             dtype: Optional[torch.dtype] = None, gated: bool = False) -> bool:
        """One layer: ``y_shard = x @ dequant(packed)^T (+ bias)``. ``consume``: x is the current exchange (pass ``x=None``);
        ``produce``: y goes to every rank's exchange buffer (and to ``out_local`` when given). ``gated`` (with ``produce``): the
        matrix interleaves this rank's gate and up rows (row ``2 r`` = gate row ``r``, row ``2 r + 1`` = up row ``r``) and the
        exchange receives ``silu(gate) * up`` - ``ns / 2`` values per rank, computed in the launch's epilogue; the down projection of
        a Llama-style FFN block then consumes it in the plain form (``parallel.ShardedFFN4bit``). Returns False - nothing launched -
        when the fused form does not serve the problem."""
        if gated and not produce:
            raise ValueError("gated=True is a form of produce=True")
        if self._broken:
            raise RuntimeError(f"this PeerChain is out of step with its peers and cannot be used any more ({self._broken}); build a new one collectively")
        st = quant_state
        ns, K = int(st.shape[0]), int(st.shape[1])
        if consume:
            if dtype is None:
                raise ValueError("consume=True needs the activation dtype")
            A = None
        else:
            if x is None or x.numel() != K or not x.is_contiguous() or x.device != self.device:
                raise ValueError("x must be one contiguous row of K values on the chain's device")
            dtype, A = x.dtype, x
        if not produce and out_local is None:
            raise ValueError("a launch that does not produce an exchange needs out_local")
        if out_local is not None and (out_local.numel() != ns or out_local.dtype != dtype or not out_local.is_contiguous()):
            raise ValueError("out_local must be a contiguous [ns] tensor of the activation dtype")
        if bias is not None and bias.dtype != dtype:
            bias = bias.to(dtype)
        if st.nested:
            absmax, absmax8, code, offset = st.state2.absmax, st.absmax, st.state2.code, st.offset
        else:
            absmax, absmax8, code, offset = st.absmax, None, None, None
        ptr = lambda t: ct.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            ok = lib.bnb_mi355x_gemv_4bit_peer(self._bufs, ct.c_void_p(self._epoch.data_ptr()), self.world, self.rank, self._dt(dtype), ptr(A), ptr(packed), ptr(absmax),
                                               ptr(absmax8), ptr(code), ptr(offset), ptr(bias), ptr(out_local), ns, K,
                                               int(st.blocksize), 1 if st.quant_type == "fp4" else 2,
                                               (1 if consume else 0) | (2 if produce else 0) | (8 if gated else 0), self.max_values, self.wg_limit,
                                               self._pending, ct.c_void_p(stream))
        if ok and produce:
            self._pending += 1
        return bool(ok)

    def read(self, n_values: int, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The current exchange - the gathered y of the last ``produce`` launch, rank-major - as a plain tensor."""
        if self._pending < 2:
            # (the region of an exchange is its position in the chain: a chain of ONE exchange would re-use its region in the very
            # next chain, before every rank has read it - csrc/gemv4_stream.hip, kChainRegions)
            raise RuntimeError("a peer chain must hold at least two exchanges before its read-out (use ShardedLinear4bit for a single layer)")
        if out is None:
            out = torch.empty(n_values, dtype=dtype, device=self.device)
        elif out.numel() != n_values or out.dtype != dtype or not out.is_contiguous():
            raise ValueError("out must be a contiguous tensor of n_values elements of dtype")
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream().cuda_stream
            lib.bnb_mi355x_peer_chain_read(self._bufs, ct.c_void_p(self._epoch.data_ptr()), self.world, self.rank, self._dt(dtype), ct.c_void_p(out.data_ptr()), int(n_values),
                                           self.max_values, self._pending, ct.c_void_p(stream))
        self._pending = 0
        return out
