#!/usr/bin/env python3
"""Backward of the 4-bit linear layer, grad_A = grad_out @ dequantize_4bit(B): the fused kernel
(bitsandbytes_amd::gemm_4bit_grad_input -> csrc/gemm4_grad_input.hip) against the reference's formulation (dequantize_4bit to
[N, K] in HBM, then a dense hipBLASLt matmul), per call over an HBM-resident rotation of layers, hipGraph-replayed."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from stream_ab import graph_time, make_layers  # noqa: E402


def main():
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--fused-only", action="store_true")
    args = ap.parse_args()
    print(torch.cuda.get_device_name(0), bnb.lib.bnb_mi355x_version().decode())
    print(f"{'N x K':>14s} {'dq':>2s} {'M':>4s} {'fused us':>9s} {'unfused us':>10s} {'(dequantize':>11s} {'+ matmul)':>9s} {'speed-up':>8s}")
    for (N, K, dq) in ((4096, 4096, False), (4096, 4096, True), (11008, 4096, False), (4096, 11008, False), (8192, 8192, False)):
        layers = make_layers(N, K, 64, "nf4", dq, cap=32)
        for M in (16, 64, 128, 256, 512):
            g = torch.randn(M, N, device="cuda").bfloat16()

            def fused():
                for q, st in layers:
                    if st.nested:
                        torch.ops.bitsandbytes_amd.gemm_4bit_grad_input.default(g, q, st.shape, st.state2.absmax, 64, "nf4",
                                                                                 st.absmax, st.state2.code, st.offset)
                    else:
                        torch.ops.bitsandbytes_amd.gemm_4bit_grad_input.default(g, q, st.shape, st.absmax, 64, "nf4")

            def deq():
                for q, st in layers:
                    F.dequantize_4bit(q, st)

            Ws = [F.dequantize_4bit(q, st) for q, st in layers[:4]]

            def mm():
                for i in range(len(layers)):
                    torch.matmul(g, Ws[i % 4])

            if M <= 256:
                tf = graph_time(fused, len(layers))
            else:
                tf = float("nan")
            if args.fused_only:
                print(f"{N:>7d}x{K:<6d} {int(dq):>2d} {M:>4d} {tf:9.2f}", flush=True)
                del Ws
                continue
            td, tm = graph_time(deq, len(layers)), graph_time(mm, len(layers))
            print(f"{N:>7d}x{K:<6d} {int(dq):>2d} {M:>4d} {tf:9.2f} {td + tm:10.2f} {td:11.2f} {tm:9.2f} {(td + tm) / tf:8.2f}", flush=True)
            del Ws
        del layers


if __name__ == "__main__":
    main()
