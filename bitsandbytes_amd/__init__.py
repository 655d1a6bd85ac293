"""bitsandbytes_amd — MI355X (gfx950) native backend for the bitsandbytes 4-bit quantized-linear path.

Public surface mirrors the reference for this path only:
``bitsandbytes_amd.functional`` (quantize_4bit, dequantize_4bit, gemv_4bit, QuantState, ...),
``bitsandbytes_amd.matmul_4bit``, ``bitsandbytes_amd.nn.{Linear4bit, LinearNF4, LinearFP4, Params4bit}``
and the ``torch.ops.bitsandbytes.*`` operators, whose HIP-device kernels call the C ABI of
``libbitsandbytes_mi355x.so`` (``include/bnb_mi355x.h``). There is no CPU implementation here by
design: without the HIP library every op raises.
"""
from . import _ops  # noqa: F401  op schemas + fake kernels (importable without a GPU)
from . import functional  # noqa: F401
from .autograd import MatMul4Bit, matmul_4bit, matmul_4bit_grouped
from .cextension import lib
from .backends import hip as _hip_backend  # noqa: F401  registers the "cuda"-key (HIP) kernels
from . import nn  # noqa: F401
from . import utils  # noqa: F401
from .parallel import (GraphedBlock, ShardedFFN4bit, ShardedLinear4bit, ShardedLinear4bitChain, ShardedLinear4bitGroup,  # noqa: F401
                       shard_ffn4bit, shard_linear4bit)

__version__ = "0.1.0"

__all__ = ["functional", "nn", "utils", "matmul_4bit", "matmul_4bit_grouped", "MatMul4Bit", "lib", "ShardedLinear4bit", "ShardedLinear4bitGroup", "ShardedLinear4bitChain", "ShardedFFN4bit", "GraphedBlock", "shard_linear4bit", "shard_ffn4bit"]
