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
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist
from torch import nn

from . import functional as F
from .autograd import matmul_4bit
from .functional import QuantState


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
                 always_gather: bool = False):
        super().__init__()
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
        buf = torch.empty((G * m, ns), dtype=y2.dtype, device=y2.device)  # rank-major concatenation
        dist.all_gather_into_tensor(buf, y2, group=self.group)
        if m == 1:
            return buf.view(*lead, G * ns)  # M == 1: rank-major already is feature-major
        return buf.view(G, m, ns).permute(1, 0, 2).reshape(*lead, G * ns)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = self.local_forward(x)
        return self.gather(y) if self.gather_output else y


def shard_linear4bit(layer, rank: Optional[int] = None, world_size: Optional[int] = None, group=None,
                     gather_output: bool = True, always_gather: bool = False) -> ShardedLinear4bit:
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
                             always_gather=always_gather)
