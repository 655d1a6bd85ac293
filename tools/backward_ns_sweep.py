#!/usr/bin/env python3
"""Fused backward: time per call over the number of N slices (bnb_mi355x_set_tuning reserved1; 0 = the built-in plan: one
workgroup per CU), HBM-resident rotation of layers, hipGraph-replayed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, bitsandbytes_amd as bnb
from stream_ab import graph_time, make_layers
for (N, K) in ((4096, 4096), (11008, 4096), (4096, 11008), (8192, 8192)):
    layers = make_layers(N, K, 64, "nf4", False, cap=32)
    for M in (16, 64, 128):
        g = torch.randn(M, N, device="cuda").bfloat16()
        row = []
        for ns in (0, 1, 2, 4, 8, 16, 32):
            bnb.lib.bnb_mi355x_set_tuning(0, ns, 0, 0)
            def fused():
                for q, st in layers:
                    torch.ops.bitsandbytes_amd.gemm_4bit_grad_input.default(g, q, st.shape, st.absmax, 64, "nf4")
            try:
                row.append(graph_time(fused, len(layers)))
            except Exception as e:
                row.append(float("nan"))
        bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
        print(f"{N}x{K} M={M:3d}  ns=auto/1/2/4/8/16/32: " + " ".join(f"{v:7.2f}" for v in row), flush=True)
    del layers
