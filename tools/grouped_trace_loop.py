#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats -- python tools/grouped_trace_loop.py [M]`: 200 grouped calls (4 x 4096^2 NF4 bs 64 bf16 members
that share x) over a rotation of 8 distinct groups (> 256 MiB): the per-kernel average is the GPU time of ONE grouped launch of the streaming MFMA
kernel (M >= 2) or of the streaming kernel (M = 1), dispatches serialised by the profiler."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bitsandbytes_amd as bnb
import bitsandbytes_amd.functional as F

M = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = K = 4096
gen = torch.Generator(device="cuda").manual_seed(0)
groups = []
for _ in range(8):
    qs, sts = [], []
    for _ in range(4):
        W = (torch.randn(N, K, device="cuda", generator=gen) / K**0.5).bfloat16()
        q, st = F.quantize_4bit(W, blocksize=64, quant_type="nf4")
        qs.append(q)
        sts.append(st)
    groups.append((qs, sts))
x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
torch.cuda.synchronize()
for i in range(200):
    qs, sts = groups[i % 8]
    bnb.matmul_4bit_grouped(x, qs, sts)
torch.cuda.synchronize()
