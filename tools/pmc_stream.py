#!/usr/bin/env python3
"""Workload for `rocprofv3 --pmc ... -- python tools/pmc_stream.py`: a few launches of the fused op at one shape over an
HBM-resident rotation of layers, so that the counters of the streaming kernel can be read from the counter_collection CSV."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=28672)
ap.add_argument("--k", type=int, default=8192)
ap.add_argument("--m", type=int, default=1)
ap.add_argument("--layers", type=int, default=4)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--backward", action="store_true", help="launch the fused backward (gemm_4bit_grad_input) with M gradient rows instead")
a = ap.parse_args()
g = torch.Generator(device="cuda").manual_seed(0)
layers = []
for _ in range(a.layers):
    W = (torch.randn(a.n, a.k, device="cuda", generator=g) / a.k**0.5).bfloat16()
    layers.append(F.quantize_4bit(W, quant_type="nf4"))
    del W
x = torch.randn(a.m, a.k, device="cuda", generator=g).bfloat16()
go = torch.randn(a.m, a.n, device="cuda", generator=g).bfloat16()
for _ in range(a.rounds):
    for q, st in layers:
        if a.backward:
            torch.ops.bitsandbytes_amd.gemm_4bit_grad_input.default(go, q, st.shape, st.absmax, 64, "nf4")
        else:
            bnb.matmul_4bit(x, q, st)
torch.cuda.synchronize()
