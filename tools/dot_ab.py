#!/usr/bin/env python3
"""A/B of the dot kernel's activation path: LDS image filled by one DMA copy per workgroup (default) vs
per-wavefront global loads (debug flag 128), production geometry, graph-replayed over 64 HBM-resident layers.
    python tools/dot_ab.py [--n 4096 --k 4096]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from bitsandbytes_amd.backends import hip  # noqa: E402


def bytes_alg(M, N, K, bs):
    return N * K // 2 + 4 * N * K // bs + 2 * M * K + 2 * M * N


def measure(layers, x, kernel, reps=10):
    L = len(layers)
    M, N = x.shape[0], layers[0][1].shape[0]
    outs = torch.empty(L, M, N, device="cuda", dtype=x.dtype)

    def step(i):
        q, st = layers[i]
        if st.nested:
            hip._gemm_4bit_fused(x, q, st.shape, st.state2.absmax, st.blocksize, st.quant_type, None, st.absmax,
                                 st.state2.code, st.offset, kernel=kernel, out=outs[i])
        else:
            hip._gemm_4bit_fused(x, q, st.shape, st.absmax, st.blocksize, st.quant_type, None, None, None, None,
                                 kernel=kernel, out=outs[i])

    for i in range(L):
        step(i)
    torch.cuda.synchronize()
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
    return e0.elapsed_time(e1) / (reps * L) * 1e3, 0.0


def run_one(x, layer, kernel=1):
    q, st = layer
    if st.nested:
        return hip._gemm_4bit_fused(x, q, st.shape, st.state2.absmax, st.blocksize, st.quant_type, None, st.absmax,
                                    st.state2.code, st.offset, kernel=kernel)
    return hip._gemm_4bit_fused(x, q, st.shape, st.absmax, st.blocksize, st.quant_type, None, None, None, None,
                                kernel=kernel)


ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--k", type=int, default=4096)
ap.add_argument("--dq", action="store_true")
ap.add_argument("--quick", action="store_true")
ap.add_argument("--m34", action="store_true")
ap.add_argument("--ldspad", action="store_true", help="M = 1 time vs extra (unused) dynamic LDS per workgroup")
ap.add_argument("--diag", action="store_true", help="EXPERIMENTAL diagonal-MFMA decode (debug flag 16) vs production, M = 1..8")
ap.add_argument("--ablate", action="store_true", help="ablations of the M = 1 kernel (results wrong by design), HBM-resident and cache-hot")
a = ap.parse_args()
N, K = a.n, a.k
L = max(8, int(640e6 // (N * K // 2)))
g = torch.Generator(device="cuda").manual_seed(0)
layers = []
for _ in range(L):
    W = (torch.randn(N, K, device="cuda", generator=g) / K**0.5).bfloat16()
    layers.append(F.quantize_4bit(W, blocksize=64, quant_type="nf4", compress_statistics=a.dq))
    del W
print(f"# N={N} K={K} dq={a.dq} layers={L}")
if a.ldspad:
    x = torch.randn(1, K, device="cuda", generator=g).bfloat16()
    print("extra LDS KiB per workgroup (on top of 32 KiB table + 8 KiB activations) -> us per launch")
    for pad in (0, 8, 16, 24, 32, 36, 40, 44, 48, 64, 80):
        bnb.lib.bnb_mi355x_set_debug(0, pad << 8)
        t, _ = measure(layers, x, 1)
        print(f"  +{pad:3d} KiB  (total {40 + pad:3d} KiB)  {t:6.2f} us")
    bnb.lib.bnb_mi355x_set_debug(0, 0)
    sys.exit(0)
if a.ablate:
    x = torch.randn(1, K, device="cuda", generator=g).bfloat16()
    names = {5: "empty (x DMA + table loads only)", 4: "weights only (no x, no absmax, no decode)",
             1: "stream only (all loads, no decode)", 2: "no table build", 3: "no weight loads", 0: "full kernel"}
    print(f"{'variant':44s} {'HBM-resident us':>16s} {'cache-hot us':>13s}")
    for abl in (5, 4, 1, 2, 3, 0):
        bnb.lib.bnb_mi355x_set_debug(abl, 0)
        t_cold, _ = measure(layers, x, 1)
        t_hot, _ = measure(layers[:2] * 16, x, 1)
        print(f"{names[abl]:44s} {t_cold:16.2f} {t_hot:13.2f}")
    bnb.lib.bnb_mi355x_set_debug(0, 0)
    sys.exit(0)
if a.diag:
    # EXPERIMENTAL diagonal-MFMA decode (debug flag 16) against the production v_dot2c decode, dot kernel forced
    # (kernel = 1) for every M; relative error of each against fp32 dequantize + matmul on layer 0
    q0, st0 = layers[0]
    W0 = F.dequantize_4bit(q0, st0).float()
    print(f"{'M':>3s} {'decode':>10s} {'us/launch':>10s} {'GB/s':>9s} {'rel err':>9s}")
    for M in (1, 2, 3, 4, 8):
        x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
        y_ref = x.float() @ W0.t()
        for flags, name in ((0, "v_dot2c"), (16, "diag-mfma"), (16 | 32, "diag lut32")):
            bnb.lib.bnb_mi355x_set_debug(0, flags)
            tg, _ = measure(layers, x, 1)
            y = run_one(x, layers[0])
            err = float((y.float() - y_ref).norm() / y_ref.norm())
            print(f"{M:3d} {name:>10s} {tg:10.2f} {bytes_alg(M, N, K, 64) / tg / 1e3:9.1f} {err:9.2e}")
    bnb.lib.bnb_mi355x_set_debug(0, 0)
    sys.exit(0)
for M in (() if a.m34 else (1, 2)):
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    for rep in range(2):
        for flags, name in ((0, "x via LDS-DMA"), (32, "x LDS, 32-copy LUT"), (128, "x per wavefront")):
            bnb.lib.bnb_mi355x_set_debug(0, flags)
            tg, te = measure(layers, x, 1)
            print(f"M={M} {name:16s} {tg:7.2f} us/launch  {bytes_alg(M, N, K, 64) / tg / 1e3:8.1f} GB/s")
# rows per wavefront x workgroup size x activation path
for M in ((3, 4) if a.m34 else (1, 2)):
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    for rpw in (1, 2):
        for wg, wname in ((64, "256thr"), (0, "512thr")):
            if a.quick and wg != 0:
                continue
            for xl, xname in ((0, "xLDS"), (128, "xwave")):
                bnb.lib.bnb_mi355x_set_tuning(rpw, 2, 0, 0)
                bnb.lib.bnb_mi355x_set_debug(0, wg | xl)
                tg, te = measure(layers, x, 1)
                print(f"M={M} rpw{rpw} {wname:7s} {xname}: {tg:7.2f} us/launch  {bytes_alg(M, N, K, 64) / tg / 1e3:8.1f} GB/s")
bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
bnb.lib.bnb_mi355x_set_debug(0, 0)
