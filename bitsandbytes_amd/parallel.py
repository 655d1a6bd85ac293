"""Output-feature (N) sharding of one ``Linear4bit`` across the GPUs of a node, with a single
all-gather of ``y`` — new functionality on top of the reference, which has no collective code at all
(SURVEY §2.1, §8e).

Why it shards: output feature ``n`` depends only on row ``n`` of the packed weight and on that row's
``K / blocksize`` absmax entries, and rows are contiguous in both buffers because quantisation
blocks run along K (requires ``K % blocksize == 0``). Rank ``r`` of ``G`` owns rows
``[r*N/G, (r+1)*N/G)``; ``x`` is replicated (it is ``M*K`` elements); every rank runs the ordinary
single-GPU kernel on its shard; ``torch.distributed.all_gather_into_tensor`` (RCCL over xGMI when
the backend is ``"nccl"``, gloo in the CPU tests) reassembles ``y[M, N]``. Nothing is reduced — K is
never split across ranks.

One process per GPU; the process group is whatever the caller initialised.

For a tensor-parallel decode (every shard launch is 3 - 4 us, every gather 2.7 KB per rank - boundaries and
latency, not bandwidth) three more pieces: ``peer=`` (a :class:`bitsandbytes_amd.peer.PeerAllGather`: the gather as ONE kernel
over peer-mapped buffers instead of a ring collective), :class:`ShardedLinear4bitGroup` (layers that share ``x`` - Q/K/V,
gate/up: one grouped launch and ONE gather for the whole group) and :class:`GraphedBlock` (a whole block of such calls captured
in one hipGraph per rank and replayed per token); and :class:`ShardedLinear4bitChain` (consecutive layers whose all-gathers are
fused into the gemv launches themselves - ``peer.PeerChain``: one launch per layer, the exchange under the next layer's weight stream).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist
from torch import nn

from . import functional as F
from .autograd import matmul_4bit, matmul_4bit_grouped
from .functional import QuantState


def _all_gather_rows(y2: torch.Tensor, world: int, group, peer) -> torch.Tensor:
    """``[m, ns] -> [G * m, ns]`` rank-major: the one-shot peer kernel when given and the message fits, else the group's own."""
    if peer is not None and y2.numel() * y2.element_size() <= peer.max_bytes:
        return peer.all_gather(y2)
    buf = torch.empty((world * y2.shape[0], y2.shape[1]), dtype=y2.dtype, device=y2.device)
    dist.all_gather_into_tensor(buf, y2, group=group)
    return buf


def shard_quant_state(packed: torch.Tensor, state: QuantState, rank: int, world_size: int):
    """Slice a packed ``[N, K]`` 4-bit weight and its QuantState to the rows owned by ``rank``.
    Returns ``(packed_shard uint8 [Ns*K/2, 1], QuantState with shape [Ns, K])``."""
    N, K = int(state.shape[0]), int(state.shape[1])
    bs = state.blocksize
    if N % world_size:
        raise ValueError(f"out_features ({N}) must be divisible by the number of ranks ({world_size})")
    if K % bs:
        raise ValueError(f"in_features ({K}) must be a multiple of blocksize ({bs}) to shard by rows")
    if (N // world_size) * K % 2:
        raise ValueError("a shard must start on a byte boundary of the packed weight")
    ns = N // world_size
    flat = packed.reshape(-1)
    if flat.dtype != torch.uint8:
        flat = flat.view(torch.uint8)
    bytes_per_shard = ns * K // 2
    packed_shard = flat[rank * bytes_per_shard : (rank + 1) * bytes_per_shard].clone().view(-1, 1)

    blocks_per_shard = ns * K // bs
    b0, b1 = rank * blocks_per_shard, (rank + 1) * blocks_per_shard
    if not state.nested:
        shard = QuantState(absmax=state.absmax[b0:b1].clone(), shape=torch.Size((ns, K)), code=state.code,
                           blocksize=bs, quant_type=state.quant_type, dtype=state.dtype)
    elif blocks_per_shard % state.state2.blocksize == 0:
        # shard boundary coincides with a boundary of the second-level blocks: slice both levels
        g = state.state2.blocksize
        s2 = QuantState(absmax=state.state2.absmax[b0 // g : b1 // g].clone(), code=state.state2.code,
                        blocksize=g, dtype=state.state2.dtype)
        shard = QuantState(absmax=state.absmax[b0:b1].clone(), shape=torch.Size((ns, K)), code=state.code,
                           blocksize=bs, quant_type=state.quant_type, dtype=state.dtype, offset=state.offset,
                           state2=s2)
    else:
        # boundary falls inside a second-level block: carry this shard's absmax un-nested (fp32);
        # the values are exactly the ones the nested form decodes to
        full = F.dequantize_blockwise(state.absmax, state.state2) + state.offset
        shard = QuantState(absmax=full.float()[b0:b1].clone(), shape=torch.Size((ns, K)), code=state.code,
                           blocksize=bs, quant_type=state.quant_type, dtype=state.dtype)
    return packed_shard, shard


class ShardedLinear4bit(nn.Module):
    """This rank's row-shard of a 4-bit linear layer; ``forward`` returns the full ``[*, N]`` output."""

    def __init__(self, packed_shard: torch.Tensor, quant_state: QuantState, out_features: int,
                 bias_shard: Optional[torch.Tensor] = None, group=None, gather_output: bool = True,
                 always_gather: bool = False, peer=None):
        super().__init__()
        # peer: a bitsandbytes_amd.peer.PeerAllGather of the same group - the gather of a decode-sized output is then one
        # kernel over peer-mapped buffers (messages above its max_bytes still take the group's all-gather)
        self.peer = peer
        # always_gather: issue the collective even in a group of one (a single-GPU smoke test of the RCCL path)
        self.always_gather = always_gather
        self.register_buffer("weight", packed_shard, persistent=False)
        self.quant_state = quant_state
        self.bias = bias_shard
        self.out_features = out_features
        self.group = group
        self.gather_output = gather_output

    @property
    def world_size(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def local_forward(self, x: torch.Tensor) -> torch.Tensor:
        bias = self.bias
        if bias is not None and bias.dtype != x.dtype:  # Linear4bit.forward casts its bias the same way
            bias = bias.to(x.dtype)
        return matmul_4bit(x, self.weight, bias=bias, quant_state=self.quant_state)

    def gather(self, y_local: torch.Tensor) -> torch.Tensor:
        """One all-gather: rank-major buffer [G, M, N/G] -> [*, N]."""
        G = self.world_size
        if G == 1 and not (self.always_gather and dist.is_initialized()):
            return y_local
        lead = y_local.shape[:-1]
        ns = y_local.shape[-1]
        y2 = y_local.reshape(-1, ns).contiguous()
        m = y2.shape[0]
        buf = _all_gather_rows(y2, G, self.group, self.peer)  # rank-major concatenation
        if m == 1:
            return buf.view(*lead, G * ns)  # M == 1: rank-major already is feature-major
        return buf.view(G, m, ns).permute(1, 0, 2).reshape(*lead, G * ns)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = self.local_forward(x)
        return self.gather(y) if self.gather_output else y


def shard_linear4bit(layer, rank: Optional[int] = None, world_size: Optional[int] = None, group=None,
                     gather_output: bool = True, always_gather: bool = False, peer=None) -> ShardedLinear4bit:
    """Build this rank's :class:`ShardedLinear4bit` from an already-quantised ``Linear4bit``."""
    if rank is None:
        rank = dist.get_rank(group)
    if world_size is None:
        world_size = dist.get_world_size(group)
    state = layer.weight.quant_state
    if state is None:
        raise ValueError("layer is not quantised yet: move it to the device first")
    packed_shard, shard_state = shard_quant_state(layer.weight.data, state, rank, world_size)
    N = int(state.shape[0])
    ns = N // world_size
    bias = None
    if layer.bias is not None:
        bias = layer.bias.data[rank * ns : (rank + 1) * ns].clone()
    return ShardedLinear4bit(packed_shard, shard_state, N, bias, group=group, gather_output=gather_output,
                             always_gather=always_gather, peer=peer)


class ShardedLinear4bitGroup(nn.Module):
    """Row shards of several 4-bit linear layers that consume the SAME input (Q/K/V, gate/up): ``forward(x)`` returns the list of
    full outputs, from ONE grouped launch (``matmul_4bit_grouped``: one kernel boundary for a decode-sized batch) writing straight
    into one communication buffer and ONE all-gather of that buffer - instead of a launch and a collective per layer.
    Values are those of the members called one by one."""

    def __init__(self, shards, always_gather: bool = False):
        super().__init__()
        shards = list(shards)
        if not shards:
            raise ValueError("empty group")
        self.shards = nn.ModuleList(shards)
        self.group = shards[0].group
        self.peer = shards[0].peer
        self.always_gather = always_gather or any(s.always_gather for s in shards)
        if any(s.group is not self.group for s in shards):
            raise ValueError("all members must live in the same process group")

    @property
    def world_size(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def forward(self, x: torch.Tensor):
        shards = list(self.shards)
        lead = x.shape[:-1]
        m = x.numel() // x.shape[-1] if x.shape[-1] else 0
        ns = [int(s.quant_state.shape[0]) for s in shards]
        # one buffer, member blocks [m, ns_i] one after the other: every block is a dense result tensor for the launch
        flat = torch.empty((1, m * sum(ns)), dtype=x.dtype, device=x.device)
        outs, off = [], 0
        for n in ns:
            outs.append(flat[0, off * m : (off + n) * m].view(m, n))
            off += n
        biases = [None if s.bias is None else (s.bias if s.bias.dtype == x.dtype else s.bias.to(x.dtype)) for s in shards]
        matmul_4bit_grouped(x.reshape(m, -1), [s.weight for s in shards], [s.quant_state for s in shards], biases, outs=outs)
        G = self.world_size
        if G == 1 and not (self.always_gather and dist.is_initialized()):
            return [o.view(*lead, n) for o, n in zip(outs, ns)]
        buf = _all_gather_rows(flat, G, self.group, self.peer)  # [G, m * sum(ns)]
        res, off = [], 0
        for n in ns:
            blk = buf[:, off * m : (off + n) * m]  # rank g's [m, n] block of this member
            if m == 1:
                res.append(blk.reshape(*lead, G * n))  # rank-major is feature-major
            else:
                res.append(blk.reshape(G, m, n).permute(1, 0, 2).reshape(*lead, G * n))
            off += n
        return res


class ShardedLinear4bitChain(nn.Module):
    """Consecutive N-sharded layers of a decode step - each one's gathered ``y`` is the next one's ``x`` (up / down projection of
    an MLP, ``out_features[i] == in_features[i + 1]``) - with every all-gather FUSED into the launches on either side of it
    (:class:`bitsandbytes_amd.peer.PeerChain`): layer ``i`` stores its shard outputs straight into every rank's exchange buffer,
    layer ``i + 1`` takes its ``x`` from there behind its own weight requests. One launch per layer + one small read-out at the
    end, instead of a kernel and a collective per layer; values are bit-identical to calling the layers one by one.

    The fused form serves one activation row (M = 1) of fp16 / bf16 and shapes within ``PeerChain.serves``; anything else -
    decided from shapes alone, so every rank decides the same - runs the members one by one (``ShardedLinear4bit.forward``).
    An activation function between two layers breaks the chain there (it needs the plain tensor): build one chain per stretch."""

    def __init__(self, shards, chain):
        # (All launches of a chain go to ONE stream. Two alternating streams - the exchange is the only real dependency between
        # consecutive layers, so layer i + 1's workgroups could take the CUs layer i's leave one by one - were tried in round 4 and
        # dead-locked: nothing orders the DISPATCH of two queues, a later launch can occupy the device before the launch it waits
        # for has been placed. DESIGN.md 6b.)
        super().__init__()
        shards = list(shards)
        if not shards:
            raise ValueError("empty chain")
        for a, b in zip(shards, shards[1:]):
            if a.out_features != int(b.quant_state.shape[1]):
                raise ValueError(f"layer outputs {a.out_features} features, the next one takes {int(b.quant_state.shape[1])}")
        self.shards = nn.ModuleList(shards)
        self.chain = chain
        self._bias_cache = {}

    def fused(self, x: torch.Tensor) -> bool:
        """Whether this call takes the fused form. Everything here is the same on every rank of a correctly used chain (shapes,
        dtypes, the autograd mode of the call, the alignment of tensors every rank built the same way) - the ranks must agree,
        because the two forms issue different collectives."""
        if self.chain is None or len(self.shards) < 2 or x.dtype not in (torch.float16, torch.bfloat16) or x.numel() != x.shape[-1]:
            return False
        # The fused launches are plain kernels: they record no autograd graph. matmul_4bit - the member-by-member path - is
        # differentiable, so a call that wants gradients must take it (dropping them silently by shape and dtype would be a bug).
        if torch.is_grad_enabled() and (x.requires_grad or any(s.bias is not None and s.bias.requires_grad for s in self.shards)):
            return False
        if x.device != self.chain.device:
            return False
        # the launcher's pointer preconditions (16-byte aligned packed weights and first-layer x), decided BEFORE any launch
        if any(s.weight.data_ptr() % 16 for s in self.shards):
            return False
        return all(self.chain.serves(int(s.quant_state.shape[0]), int(s.quant_state.shape[1]), int(s.quant_state.blocksize), i > 0)
                   and self.chain.world * int(s.quant_state.shape[0]) == s.out_features
                   for i, s in enumerate(self.shards))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self.fused(x):
            for s in self.shards:
                x = s(x)
            return x
        lead = x.shape[:-1]
        x1 = x.reshape(-1).contiguous()
        if x1.data_ptr() % 16:
            x1 = x1.clone()  # (a view at an odd offset: the launcher wants 16-byte aligned activations)
        biases = []
        for i, s in enumerate(self.shards):  # everything that can raise on the host happens before the first launch
            bias = s.bias
            if bias is not None and bias.dtype != x.dtype:
                key = (i, x.dtype, bias.data_ptr(), bias._version)
                if self._bias_cache.get("key" + str(i)) != key:
                    self._bias_cache["key" + str(i)], self._bias_cache[i] = key, bias.to(x.dtype)
                bias = self._bias_cache[i]
            biases.append(bias)
        done = 0
        try:
            for i, s in enumerate(self.shards):
                if not self.chain.gemv(x1 if i == 0 else None, s.weight, s.quant_state, bias=biases[i], consume=i > 0, produce=True, dtype=x.dtype):
                    raise RuntimeError(f"PeerChain refused layer {i} although its own shape check accepted it")
                done += 1
            return self.chain.read(self.shards[-1].out_features, x.dtype).view(*lead, self.shards[-1].out_features)
        except Exception as exc:
            if done:
                # Some exchanges of this chain are out and its read-out is not: this rank's count of pending exchanges no longer
                # matches its peers', and every later call would consume the wrong regions. The chain object is poisoned - every
                # further use raises at once instead of desynchronising silently; a new PeerChain has to be built collectively.
                self.chain._broken = f"a chain of {len(self.shards)} layers stopped after {done}: {type(exc).__name__}: {exc}"
            raise


def _plain_absmax(state: QuantState) -> torch.Tensor:
    """fp32 absmax of a (possibly double-quantised) state: exactly the values the nested kernels reconstruct (two roundings:
    the product code2[q8] * absmax2 and the sum with the offset - what `dequantize_blockwise` + a torch add compute)."""
    if not state.nested:
        return state.absmax.float()
    return (F.dequantize_blockwise(state.absmax, state.state2) + state.offset).float()


class ShardedFFN4bit(nn.Module):
    """One gated FFN block  ``y = down(act(gate(x)) * up(x))``  (Llama: ``act`` = SiLU) of a decode step, every projection N-sharded
    over the ranks - BASELINE.json ``configs[3]`` as it states it: the Llama FFN matrices sharded across the GPUs of a node. With a
    :class:`bitsandbytes_amd.peer.PeerChain` the block is TWO launches and one small read-out per rank and token:

    1. one launch over this rank's gate and up rows, INTERLEAVED into one ``[2 ns, H]`` matrix at construction (row ``2 r`` = gate row
       ``r``, row ``2 r + 1`` = up row ``r``: they share ``x``, and the thread pair that exchanges its outputs in the kernel's epilogue
       then holds ``g`` and ``u`` of one activation). The epilogue computes ``silu(g) * u`` (``gemv(..., gated=True)``: each op in
       fp32, rounded once to the 16-bit type - torch's arithmetic for ``F.silu(g) * u``) and stores THAT into every rank's exchange
       buffer: ``ns`` values per rank instead of ``2 ns``;
    2. the down shard's launch, which takes its input from that exchange in the chain's plain form, outputs to the next exchange;
    3. ``read``: the gathered ``y``.

    Values are bit-identical to the unsharded block (``down(F.silu(gate(x)) * up(x))`` with the three ``Linear4bit`` layers) and to
    the member-by-member path, which every call outside the fused form takes (more than one activation row, fp32, gradients, shapes
    ``PeerChain.serves`` refuses): a grouped launch for gate / up, one gather, torch's activation, the down shard, one gather.
    Decided from shapes, dtypes and the call's autograd mode - the same on every rank. Memory: the interleaved [gate; up] matrix is a
    second copy of the two shards (the member-by-member path keeps using the members' own buffers) - 2 x F/G x H / 2 bytes per rank and
    block, 7.3 MB for an 8-way shard of Llama-3-8B's FFN. Nothing in the reference to mirror (it has no
    collective code); the block's arithmetic is the reference's ``Linear4bit`` x 3 (nn/modules.py:609-637)."""

    def __init__(self, gate: ShardedLinear4bit, up: ShardedLinear4bit, down: ShardedLinear4bit, chain=None):
        super().__init__()
        sg, su = gate.quant_state, up.quant_state
        if tuple(sg.shape) != tuple(su.shape) or sg.blocksize != su.blocksize or sg.quant_type != su.quant_type:
            raise ValueError("gate and up must have the same shard shape, blocksize and quant_type")
        if gate.out_features != up.out_features or gate.out_features != int(down.quant_state.shape[1]):
            raise ValueError("down must take what gate / up produce")
        self.gate, self.up, self.down = gate, up, down
        self.group = ShardedLinear4bitGroup([gate, up])
        self.chain = chain
        ns, H = int(sg.shape[0]), int(sg.shape[1])
        # ONE matrix with the rows interleaved (g0, u0, g1, u1, ...): packed bytes and absmax are row-major over [N, K] with whole
        # rows (K % blocksize == 0, K even), so interleaving the two shards' buffers row by row IS the packed form of that matrix.
        # Nested statistics are carried un-nested (the two matrices have their own offsets and second-level tables; the fp32 values
        # are exactly the ones the nested kernels reconstruct).
        wg, wu = gate.weight.reshape(-1), up.weight.reshape(-1)
        if wg.dtype != torch.uint8:
            wg, wu = wg.view(torch.uint8), wu.view(torch.uint8)
        self.register_buffer("gu_weight", torch.stack([wg.view(ns, -1), wu.view(ns, -1)], dim=1).reshape(-1, 1).contiguous(), persistent=False)
        am = torch.stack([_plain_absmax(sg).view(ns, -1), _plain_absmax(su).view(ns, -1)], dim=1).reshape(-1).contiguous()
        self.gu_state = QuantState(absmax=am, shape=torch.Size((2 * ns, H)), code=sg.code, blocksize=sg.blocksize, quant_type=sg.quant_type, dtype=sg.dtype)
        self._gu_bias = {}

    def _stacked_bias(self, dtype):
        bg, bu = self.gate.bias, self.up.bias
        if bg is None and bu is None:
            return None
        key = (dtype, None if bg is None else (bg.data_ptr(), bg._version), None if bu is None else (bu.data_ptr(), bu._version))
        if self._gu_bias.get("key") != key:
            ns = int(self.gate.quant_state.shape[0])
            dev = self.gu_weight.device
            parts = [(b.to(dtype) if b is not None else torch.zeros(ns, dtype=dtype, device=dev)) for b in (bg, bu)]
            self._gu_bias = {"key": key, "bias": torch.stack(parts, dim=1).reshape(-1).contiguous()}  # interleaved like the rows
        return self._gu_bias["bias"]

    def fused(self, x: torch.Tensor) -> bool:
        chain = self.chain
        if chain is None or x.dtype not in (torch.float16, torch.bfloat16) or x.numel() != x.shape[-1] or x.device != chain.device:
            return False
        members = (self.gate, self.up, self.down)
        if torch.is_grad_enabled() and (x.requires_grad or any(m.bias is not None and m.bias.requires_grad for m in members)):
            return False  # (the fused launches record no autograd graph: see ShardedLinear4bitChain.fused)
        if self.gu_weight.data_ptr() % 16 or self.down.weight.data_ptr() % 16:
            return False
        sd = self.down.quant_state
        ns, H = int(self.gate.quant_state.shape[0]), int(self.gate.quant_state.shape[1])
        Fdim = int(sd.shape[1])
        return (chain.world * ns == Fdim == self.gate.out_features and chain.world * int(sd.shape[0]) == self.down.out_features
                and chain.serves(2 * ns, H, int(self.gu_state.blocksize), consume=False, gated=True)
                and chain.serves(int(sd.shape[0]), Fdim, int(sd.blocksize), consume=True))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self.fused(x):
            g, u = self.group(x)
            return self.down(torch.nn.functional.silu(g) * u)
        lead = x.shape[:-1]
        x1 = x.reshape(-1).contiguous()
        if x1.data_ptr() % 16:
            x1 = x1.clone()
        gu_bias = self._stacked_bias(x.dtype)  # (everything that can raise on the host happens before the first launch)
        db = self.down.bias
        if db is not None and db.dtype != x.dtype:
            db = db.to(x.dtype)
        chain, done = self.chain, 0
        try:
            if not chain.gemv(x1, self.gu_weight, self.gu_state, bias=gu_bias, consume=False, produce=True, gated=True):
                raise RuntimeError("PeerChain refused the gate / up launch although its own shape check accepted it")
            done = 1
            if not chain.gemv(None, self.down.weight, self.down.quant_state, bias=db, consume=True, produce=True, dtype=x.dtype):
                raise RuntimeError("PeerChain refused the down launch although its own shape check accepted it")
            done = 2
            return chain.read(self.down.out_features, x.dtype).view(*lead, self.down.out_features)
        except Exception as exc:
            if done:
                chain._broken = f"an FFN block stopped after {done} of its 2 launches: {type(exc).__name__}: {exc}"
            raise


def shard_ffn4bit(gate, up, down, rank: Optional[int] = None, world_size: Optional[int] = None, group=None, chain=None, peer=None) -> ShardedFFN4bit:
    """This rank's :class:`ShardedFFN4bit` from three already-quantised ``Linear4bit`` layers (gate, up: ``H -> F``; down: ``F -> H``)."""
    members = [shard_linear4bit(layer, rank, world_size, group=group, peer=peer) for layer in (gate, up, down)]
    return ShardedFFN4bit(*members, chain=chain)


class GraphedBlock:
    """A block of stream-ordered work - sharded launches, grouped launches, peer gathers, anything capturable - recorded ONCE into
    a hipGraph per rank and replayed per call: the per-kernel boundary is paid by the GPU's command processor instead of by
    Python and the launch path (at 3 - 4 us per shard launch the host is the bound otherwise - SURVEY §8e).
    ``fn(*tensors) -> tensor | sequence of tensors``; the example inputs fix shapes and dtypes. Every rank must build and call its
    GraphedBlock in the same order (the collectives inside are replayed, not renegotiated). Inference only."""

    def __init__(self, fn, *example_inputs: torch.Tensor, warmup: int = 2, peers=(), check_every: int = 1024):
        # peers: the PeerAllGather objects the block uses. A peer kernel that gives up waiting marks the missing rows NaN and sets a
        # sticky status word; nothing on the device raises. The block reads the word every `check_every` replays (a device
        # synchronisation each time) and in check(): a dead rank surfaces as an exception instead of NaNs far downstream.
        self._peers = [p for p in peers if p is not None]
        self._check_every = max(1, int(check_every))
        self._replays = 0
        self._static_in = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(max(1, warmup)):  # lazy initialisation (workspaces, RCCL channels) happens outside the capture
                fn(*self._static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            out = fn(*self._static_in)
        self._single = isinstance(out, torch.Tensor)
        self._static_out = [out] if self._single else list(out)

    def __call__(self, *inputs: torch.Tensor):
        if len(inputs) != len(self._static_in):
            raise ValueError(f"expected {len(self._static_in)} inputs")
        for dst, src in zip(self._static_in, inputs):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError("GraphedBlock inputs must keep the shapes and dtypes of the example inputs")
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        self.graph.replay()
        self._replays += 1
        if self._peers and self._replays % self._check_every == 0:
            self.check()
        return self._static_out[0] if self._single else list(self._static_out)

    def check(self) -> None:
        """Raises if a peer collective of this block ever gave up waiting for a rank (synchronises the device)."""
        for p in self._peers:
            p.check()

    @property
    def inputs(self):
        """The graph's own input tensors: write into them directly to skip the copy in ``__call__``."""
        return list(self._static_in)
