#!/usr/bin/env python3
"""Eager-mode (no hipGraph) cost per call of the Python layers above the C ABI, M = 1, N = K = 4096:
Linear4bit.forward -> matmul_4bit -> torch.ops.bitsandbytes.gemm_4bit -> backend glue -> ctypes.
The GPU work is ~5 us per call, so the wall time per call IS the host overhead once the queue is full."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from bitsandbytes_amd.backends import hip  # noqa: E402

N = K = 4096
W = (torch.randn(N, K, device="cuda") / 64).bfloat16()
layer = bnb.nn.Linear4bit(K, N, bias=False, quant_type="nf4", compress_statistics=False, compute_dtype=torch.bfloat16)
layer.weight = bnb.nn.Params4bit(W, requires_grad=False, quant_type="nf4", compress_statistics=False, module=layer)
layer = layer.cuda()
q, st = layer.weight.data, layer.weight.quant_state
x = torch.randn(1, K, device="cuda", dtype=torch.bfloat16)
out = torch.empty(1, N, device="cuda", dtype=torch.bfloat16)


def timeit(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


with torch.no_grad():
    rows = [
        ("Linear4bit.forward (prepared C++ call)", lambda: layer(x)),
        ("Linear4bit.forward (Python layers)", lambda: (layer._prepared_drop(), layer(x))[1]),
        ("matmul_4bit", lambda: bnb.matmul_4bit(x, q, st)),
        ("torch.ops.bitsandbytes.gemm_4bit", lambda: torch.ops.bitsandbytes.gemm_4bit.default(x, q, st.shape, st.absmax, 64, "nf4")),
        ("Python kernel of the op (replaced)", lambda: hip._gemm_4bit_python_kernel(x, q, st.shape, st.absmax, 64, "nf4")),
        ("backend _gemm_4bit_fused(out=)", lambda: hip._gemm_4bit_fused(x, q, st.shape, st.absmax, 64, "nf4", None, None, None, None, out=out)),
        ("torch.add (reference point)", lambda: torch.add(x, x)),
    ]
    print(f"# native dispatch (csrc/torch_dispatch.cpp) loaded: {hip.NATIVE_DISPATCH}")
    for name, fn in rows:
        print(f"{name:40s} {timeit(fn):7.1f} us per call")


# ---- load time (round 5): what quantizing one 4096^2 layer costs end to end in eager mode, wall time per call with the queue full
# (the larger of host dispatch and GPU time): plain statistics, double quantisation (quantize4 + mean + subtract + the 8-bit
# quantize of 262 144 absmax values: the default of HF NF4 checkpoints' loaders), and the pieces of the latter on their own.
Wq = (torch.randn(N, K, device="cuda") / 64).bfloat16()
am = torch.rand(N * K // 64, device="cuda") + 0.5
with torch.no_grad():
    rows = [
        ("quantize_4bit NF4 bs64", lambda: F.quantize_4bit(Wq, blocksize=64, quant_type="nf4")),
        ("quantize_4bit NF4 bs64 + double quant", lambda: F.quantize_4bit(Wq, blocksize=64, quant_type="nf4", compress_statistics=True)),
        ("  absmax.mean()", lambda: am.mean()),
        ("  absmax - offset", lambda: am - 0.25),
        ("  quantize_blockwise(absmax, 256)", lambda: F.quantize_blockwise(am, blocksize=256)),
    ]
    print("# load time, 4096^2 bf16 -> NF4 (wall us per call, eager, queue full)")
    for name, fn in rows:
        print(f"{name:40s} {timeit(fn, n=500):7.1f} us per call")
