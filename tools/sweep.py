#!/usr/bin/env python3
"""GPU-side micro-benchmark sweep over the MFMA kernel geometries (cfg x K-slices) and M. Prints a table; each cell = microseconds per launch measured two ways:
  graph  : 10 replays of a hipGraph of 64 back-to-back launches over 64 distinct layers (HBM-resident rotation)
  evpair : mean of per-launch HIP-event brackets (eager)
plus the implied algorithmic GB/s from `graph`. Usage: python tools/sweep.py [--n 4096 --k 4096] [--quick]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from bitsandbytes_amd.backends import hip  # noqa: E402

L = 64


def bytes_alg(M, N, K, bs):
    return N * K // 2 + 4 * N * K // bs + 2 * M * K + 2 * M * N


CODE16 = None


def measure(layers, x, kernel, reps=10):
    M = x.shape[0]
    N = layers[0][1].shape[0]
    outs = torch.empty(L, M, N, device="cuda", dtype=x.dtype)

    def step(i):
        q, st = layers[i % L]
        hip._gemm_4bit_fused(x, q, st.shape, st.absmax, st.blocksize, st.quant_type, None, None, None, None,
                             kernel=kernel, out=outs[i % L], code16=CODE16)

    for i in range(L):
        step(i)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for i in range(L):
            step(i)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(L):
            step(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    t_graph = e0.elapsed_time(e1) / (reps * L) * 1e3
    n = 256
    st_, en_ = [torch.cuda.Event(enable_timing=True) for _ in range(n)], [torch.cuda.Event(enable_timing=True) for _ in range(n)]
    for i in range(n):
        st_[i].record()
        step(i)
        en_[i].record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in zip(st_, en_))[: int(n * 0.9)]
    return t_graph, sum(ts) / len(ts) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--k", type=int, default=4096)
    ap.add_argument("--bs", type=int, default=64)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--mfma-only", action="store_true")
    ap.add_argument("--cfgs", default="11,12,13,14,20,21,22", help="MFMA kernel geometries to sweep (cfg codes, see make_plan)")
    ap.add_argument("--ms", default="", help="comma list of M values for the MFMA sweep")
    ap.add_argument("--kss", default="2,4,8,16", help="K-slice counts to sweep")
    a = ap.parse_args()
    N, K, bs = a.n, a.k, a.bs
    g = torch.Generator(device="cuda").manual_seed(0)
    layers = []
    for _ in range(L):
        W = (torch.randn(N, K, device="cuda", generator=g) / K**0.5).bfloat16()
        layers.append(F.quantize_4bit(W, blocksize=bs, quant_type="nf4"))
        del W
    print(f"# N={N} K={K} bs={bs} layers={L} ({L * bytes_alg(1, N, K, bs) / 1e6:.0f} MB rotated)")
    print(f"{'kernel':8s} {'M':>3s} {'knobs':>12s} {'graph_us':>9s} {'evpair_us':>9s} {'GB/s(graph)':>11s} {'TFLOP/s':>8s}")
    Ms = [1, 8, 16, 64] if a.quick else [1, 4, 8, 16, 32, 64]
    if a.ms:
        Ms = [int(v) for v in a.ms.split(",")]
    names = {11: "pc8x1", 12: "pc4x2", 13: "pc8x2", 14: "pc4x1", 20: "rt", 21: "rt8w", 22: "rt16w"}
    CFG = {int(c): names.get(int(c), f"cfg{c}") for c in a.cfgs.split(",")}
    KSS = tuple(int(v) for v in a.kss.split(","))
    for M in Ms:
        x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        mt = (M + 15) // 16
        bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
        tg, te = measure(layers, x, 2)
        print(f"{'mfma':8s} {M:3d} {'auto':>14s} {tg:9.2f} {te:9.2f} {bytes_alg(M, N, K, bs) / tg / 1e3:11.1f} {2 * M * N * K / tg / 1e6:8.2f}")
        if a.quick:
            continue
        for cfg, cname in CFG.items():
            for ks in ((0, 2) if cfg >= 20 else KSS):
                bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, cfg * 100 + ks)
                tg, te = measure(layers, x, 2)
                print(f"{'mfma':8s} {M:3d} {f'{cname} ks{ks}':>14s} {tg:9.2f} {te:9.2f} {bytes_alg(M, N, K, bs) / tg / 1e3:11.1f} {2 * M * N * K / tg / 1e6:8.2f}")
    bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
    # standalone quantize / dequantize streams (C1 shapes)
    W = (torch.randn(4096, 4096, device="cuda") ).half()
    q, st = F.quantize_4bit(W, quant_type="nf4")
    for name, fn, nb in (("quant4", lambda: F.quantize_4bit(W, quant_type="nf4"), 4096 * 4096 * 2.5625),
                         ("dequant4", lambda: F.dequantize_4bit(q, st), 4096 * 4096 * 2.5625)):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 50 * 1e3
        print(f"{name:8s}   - {'4096x4096 fp16':>12s} {t:9.2f} {'':9s} {nb / t / 1e3:11.1f}")


if __name__ == "__main__":
    main()
