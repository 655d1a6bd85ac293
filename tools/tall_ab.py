#!/usr/bin/env python3
"""Round 6: the tall-tile kernel (csrc/gemm4_mfma_tall.hip, forced with knob cfg 60) against the public op's route (K-quarter
kernel in 64-row passes up to fused_max_m(), dequantize + hipBLASLt above) and against dequantize + hipBLASLt alone: us per call
over an HBM-resident rotation of layers (hipGraph, round-robin medians), relative error vs fp32 dequantize + fp64 matmul, TFLOP/s.
    python tools/tall_ab.py [--quick] [--m 128,256,512,1024,2048]"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from bitsandbytes_amd.backends import hip  # noqa: E402
from stream_ab import make_layers  # noqa: E402
from stream_prologue_ab import timed  # noqa: E402


def fused(q, st, x, kernel, out=None):
    if st.nested:
        return hip._gemm_4bit_fused(x, q, st.shape, st.state2.absmax, st.blocksize, st.quant_type, None, st.absmax, st.state2.code, st.offset,
                                    kernel=kernel, out=out)
    return hip._gemm_4bit_fused(x, q, st.shape, st.absmax, st.blocksize, st.quant_type, None, None, None, None, kernel=kernel, out=out)


def graph_of(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    return g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--m", default="128,256,512,576,768,1024,1536,2048")
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    print(torch.cuda.get_device_name(0), bnb.lib.bnb_mi355x_version().decode())
    cases = [(4096, 4096, 64, "nf4", False), (8192, 8192, 64, "nf4", False), (11008, 4096, 64, "nf4", False), (4096, 11008, 64, "nf4", True),
             (4096, 4096, 128, "fp4", True)]
    if args.quick:
        cases = cases[:2]
    print(f"{'N x K':>14s} {'bs':>4s} {'qt':>3s} {'dq':>2s} {'M':>5s} {'public op':>10s} {'tall':>8s} {'unfused':>8s}   TF/s tall (% of 2.5 PF)   err public / tall")
    for (N, K, bs, qt, dq) in cases:
        layers = make_layers(N, K, bs, qt, dq, cap=8)
        L = len(layers)
        Wd = F.dequantize_4bit(*layers[0]).double()
        for M in (int(v) for v in args.m.split(",")):
            x = torch.randn(M, K, device="cuda").bfloat16()
            outs = [torch.empty(M, N, device="cuda", dtype=torch.bfloat16) for _ in layers]

            def pub():
                for (q, st) in layers:
                    bnb.matmul_4bit(x, q, st)

            def tall():
                for (q, st), o in zip(layers, outs):
                    fused(q, st, x, 2, o)

            def unf():
                for (q, st) in layers:
                    torch.nn.functional.linear(x, F.dequantize_4bit(q, st))

            y_pub = bnb.matmul_4bit(x, *layers[0])
            bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 6000)
            y_tall = fused(*layers[0], x, 2).clone()
            fam = bnb.lib.bnb_mi355x_last_gemm_kernel()
            g_tall = graph_of(tall)
            bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
            g_pub, g_unf = graph_of(pub), graph_of(unf)
            ref = x.double() @ Wd.t()
            errs = [float((y.double() - ref).norm() / ref.norm()) for y in (y_pub, y_tall)]
            graphs = [g_pub, g_tall, g_unf]
            t0 = timed(graphs[0], L, 3)
            reps = max(3, int(12000.0 / (t0 * L)) + 1)
            samples = [[] for _ in graphs]
            for r in range(args.rounds):
                order = list(range(3))
                if r % 2:
                    order.reverse()
                for i in order:
                    samples[i].append(timed(graphs[i], L, reps))
            med = [statistics.median(s_) for s_ in samples]
            tf = 2.0 * M * N * K / med[1] / 1e6
            flag = "" if (errs[1] < 1e-2 and fam == 8) else f"   <-- FAIL (family {fam})"
            print(f"{N:>7d}x{K:<6d} {bs:>4d} {qt:>3s} {int(dq):>2d} {M:>5d} {med[0]:10.2f} {med[1]:8.2f} {med[2]:8.2f}   {tf:8.1f} ({tf / 25:.1f})   "
                  f"{errs[0]:.1e} / {errs[1]:.1e}{flag}", flush=True)
            del graphs, g_pub, g_tall, g_unf
        del layers


if __name__ == "__main__":
    main()
