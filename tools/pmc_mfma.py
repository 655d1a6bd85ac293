#!/usr/bin/env python3
"""Workload for `rocprofv3 --pmc ... -- python tools/pmc_mfma.py`: a few launches of the fused op at a given
shape so that the counters of the MFMA kernel can be read from the counter_collection CSV."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=8192)
ap.add_argument("--k", type=int, default=8192)
ap.add_argument("--m", type=int, default=64)
ap.add_argument("--layers", type=int, default=8)
ap.add_argument("--knob", type=int, default=0, help="bnb_mi355x_set_tuning mfma_knob1 (e.g. 4000 = K-quarter kernel)")
a = ap.parse_args()
g = torch.Generator(device="cuda").manual_seed(0)
layers = []
for _ in range(a.layers):
    W = (torch.randn(a.n, a.k, device="cuda", generator=g) / a.k**0.5).bfloat16()
    layers.append(F.quantize_4bit(W, quant_type="nf4"))
    del W
x = torch.randn(a.m, a.k, device="cuda", generator=g).bfloat16()
bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, a.knob)
for _ in range(3):
    for q, st in layers:
        bnb.matmul_4bit(x, q, st)
torch.cuda.synchronize()
