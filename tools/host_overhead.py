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
    # round 6: a group of four layers that share x (Q/K/V/O) in eager mode, us per GROUP - wall time with the queue full
    group = []
    for _ in range(4):
        lay = bnb.nn.Linear4bit(K, N, bias=False, quant_type="nf4", compress_statistics=False, compute_dtype=torch.bfloat16)
        lay.weight = bnb.nn.Params4bit((torch.randn(N, K, device="cuda") / 64).bfloat16(), requires_grad=False, quant_type="nf4", compress_statistics=False, module=lay)
        group.append(lay.cuda())
    qs, sts = [g.weight.data for g in group], [g.weight.quant_state for g in group]
    for m_rows in (1, 4, 16):
        xm = torch.randn(m_rows, K, device="cuda", dtype=torch.bfloat16)
        [g(xm) for g in group]  # (prepares the layers)
        print(f"# group of 4 x 4096^2, M = {m_rows}")
        for name, fn in (("  4 x Linear4bit.forward (prepared)", lambda: [g(xm) for g in group]),
                         ("  linear4bit_group_forward (prepared)", lambda: bnb.nn.linear4bit_group_forward(group, xm)),
                         ("  matmul_4bit_grouped (Python glue)", lambda: bnb.matmul_4bit_grouped(xm, qs, sts))):
            print(f"{name:40s} {timeit(fn, 1000):7.1f} us per group")


# ---- load time (round 5): what quantizing one 4096^2 layer costs end to end in eager mode, wall time per call with the queue full
# (the larger of host dispatch and GPU time): plain statistics, double quantisation (quantize4 + mean + subtract + the 8-bit
# quantize of 262 144 absmax values: the default of HF NF4 checkpoints' loaders), and the pieces of the latter on their own.
Wq = (torch.randn(N, K, device="cuda") / 64).bfloat16()
am = torch.rand(N * K // 64, device="cuda") + 0.5
with torch.no_grad():
    rows = [
        ("quantize_4bit NF4 bs64", lambda: F.quantize_4bit(Wq, blocksize=64, quant_type="nf4")),
        ("quantize_4bit NF4 bs64 + double quant", lambda: F.quantize_4bit(Wq, blocksize=64, quant_type="nf4", compress_statistics=True)),
        ("  the raw operator (4-bit encoder)", lambda: torch.ops.bitsandbytes.quantize_4bit.default(Wq, 64, "nf4", torch.uint8)),
        ("  the nested operator (3 launches)", lambda: torch.ops.bitsandbytes_amd.quantize_4bit_nested.default(Wq, F._dynamic_map(Wq.device), 64, "nf4", torch.uint8)),
        ("  get_4bit_type (copy of a constant)", lambda: F.get_4bit_type("nf4", device=Wq.device)),
        ("  (reference sequence) absmax.mean()", lambda: am.mean()),
        ("  (reference sequence) absmax - offset", lambda: am - 0.25),
        ("  (reference sequence) quantize_blockwise", lambda: F.quantize_blockwise(am, blocksize=256)),
    ]
    print("# load time, 4096^2 bf16 -> NF4 (wall us per call, eager, queue full)")
    for name, fn in rows:
        print(f"{name:40s} {timeit(fn, n=500):7.1f} us per call")


# ---- the same three sequences as GPU time (hipGraph replay, no host in the loop): what the launches themselves cost
def graph_us(fn, reps=200):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st_ = torch.cuda.Stream()
    st_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st_):
        fn()
    torch.cuda.current_stream().wait_stream(st_)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (reps * 10) * 1e3)
    return best


code8 = F._dynamic_map(Wq.device)


def reference_sequence():
    p_, a_ = torch.ops.bitsandbytes.quantize_4bit.default(Wq, 64, "nf4", torch.uint8)
    o_ = a_.mean()
    return p_, torch.ops.bitsandbytes.quantize_blockwise.default(a_ - o_, code8, 256), o_


with torch.no_grad():
    print("# GPU time of the same work (hipGraph replay of 10 calls, us per call; the input is re-read from HBM/MALL every call)")
    print(f"{'4-bit encoder alone':40s} {graph_us(lambda: torch.ops.bitsandbytes.quantize_4bit.default(Wq, 64, 'nf4', torch.uint8)):7.1f} us")
    print(f"{'nested operator (3 launches)':40s} {graph_us(lambda: torch.ops.bitsandbytes_amd.quantize_4bit_nested.default(Wq, code8, 64, 'nf4', torch.uint8)):7.1f} us")
    print(f"{'reference sequence (5 launches)':40s} {graph_us(reference_sequence):7.1f} us")


# ---- dequantize_4bit of a double-quantised state (the unfused route above 512 rows and the unfused backward call it once per layer)
with torch.no_grad():
    qn, stn = F.quantize_4bit(Wq, blocksize=64, quant_type="nf4", compress_statistics=True)

    def three_ops():
        a_ = torch.ops.bitsandbytes.dequantize_blockwise.default(stn.absmax, stn.state2.absmax, stn.state2.code, 256, torch.float32)
        a_ += stn.offset
        return torch.ops.bitsandbytes.dequantize_4bit.default(qn, a_, 64, "nf4", [N, K], torch.bfloat16)

    print("# dequantize_4bit, nested statistics, 4096^2 -> bf16: wall us per call (eager, queue full) | GPU us (hipGraph replay)")
    print(f"{'one operator / one launch':40s} {timeit(lambda: F.dequantize_4bit(qn, stn), n=500):7.1f} | {graph_us(lambda: F.dequantize_4bit(qn, stn)):6.1f}")
    print(f"{'reference sequence (three operators)':40s} {timeit(three_ops, n=500):7.1f} | {graph_us(three_ops):6.1f}")
    print(f"{'plain statistics (one operator)':40s} {timeit(lambda: F.dequantize_4bit(q, st), n=500):7.1f} | {graph_us(lambda: F.dequantize_4bit(q, st)):6.1f}")
