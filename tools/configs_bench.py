#!/usr/bin/env python3
"""Per-launch time of the fused op on every BASELINE.json config shape (single GPU), HBM-resident
rotation, hipGraph-replayed. Prints one row per (config, M): us, algorithmic GB/s, TFLOP/s, and the
fractions of the HBM (8 TB/s) and dense bf16 MFMA (2.5 PF/s) peaks."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402


def alg_bytes(M, N, K, bs, nested):
    w = N * K // 2
    s = (N * K // bs) * (1 if nested else 4) + ((4 * ((N * K // bs + 255) // 256) + 1028) if nested else 0)
    return w + s + 2 * M * K + 2 * M * N


def bench(N, K, M, qt, bs, dq, reps=5, hot=False):
    """hot=False: rotate over > 600 MB of distinct layers (every launch streams from HBM).
    hot=True: two layers only (<= 2 x 40 MB), i.e. weights resident in the 256 MiB Infinity Cache / L2."""
    per_layer = alg_bytes(1, N, K, bs, dq)
    L = 2 if hot else max(2, min(64, int(600e6 // per_layer) + 1))
    g = torch.Generator(device="cuda").manual_seed(0)
    layers = []
    for _ in range(L):
        W = (torch.randn(N, K, device="cuda", generator=g) / K**0.5).bfloat16()
        layers.append(F.quantize_4bit(W, blocksize=bs, quant_type=qt, compress_statistics=dq))
        del W
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()

    rounds = 32 if hot else 1

    def chunk():
        for _ in range(rounds):
            for j in range(L):
                q, st = layers[j]
                bnb.matmul_4bit(x, q, st)

    for _ in range(2):
        chunk()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        chunk()
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        chunk()
    gr.replay()
    torch.cuda.synchronize()
    def region(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (n * L * rounds) * 1e3

    # (round 5: a single region of 5 replays - well under a millisecond on the small shapes - carried a first-measured penalty of
    # up to 8 %, found with tools/stream_prologue_ab.py. One untimed block, then the MEDIAN of three regions of >= 10 ms each.)
    t0 = region(reps)
    n = max(reps, int(10000.0 / (t0 * L * rounds)) + 1)
    us = sorted(region(n) for _ in range(3))[1]
    b = alg_bytes(M, N, K, bs, dq)
    fl = 2 * M * N * K
    return us, b / us / 1e3, fl / us / 1e6, L


CONFIGS = [
    ("C2 gemv NF4 bs64", 4096, 4096, "nf4", 64, False, (1,)),
    ("headline sweep", 4096, 4096, "nf4", 64, False, (2, 4, 8, 16, 32, 64)),
    ("C3 gemm NF4 bs64", 8192, 8192, "nf4", 64, False, (64, 16, 1)),
    ("C4 FFN up 11008x4096", 11008, 4096, "nf4", 64, False, (1, 64)),
    ("C4 FFN down 4096x11008", 4096, 11008, "nf4", 64, False, (1, 64)),
    ("C4 per-GPU shard 1376x4096", 1376, 4096, "nf4", 64, False, (1, 64)),
    ("C4 per-GPU shard 512x11008", 512, 11008, "nf4", 64, False, (1, 64)),
    ("Llama-3 FFN 14336x4096", 14336, 4096, "nf4", 64, False, (1,)),
    ("28672x8192 (70B-class FFN)", 28672, 8192, "nf4", 64, False, (1,)),
    ("C5 FP4 DQ bs128", 4096, 4096, "fp4", 128, True, (1, 16)),
    ("NF4 DQ bs64 (Linear4bit default)", 4096, 4096, "nf4", 64, True, (1, 16)),
]
print(f"{'config':34s} {'M':>3s} {'us':>8s} {'GB/s':>8s} {'%HBM':>6s} {'TFLOP/s':>8s} {'%MFMA':>6s} {'layers':>6s} {'cache-hot us':>12s}")
for name, N, K, qt, bs, dq, Ms in CONFIGS:
    for M in Ms:
        us, gbs, tf, L = bench(N, K, M, qt, bs, dq)
        us_hot = bench(N, K, M, qt, bs, dq, hot=True)[0]
        print(f"{name:34s} {M:3d} {us:8.2f} {gbs:8.1f} {gbs / 80:6.1f} {tf:8.2f} {tf / 25:6.2f} {L:6d} {us_hot:12.2f}", flush=True)
