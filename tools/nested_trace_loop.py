#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats -- python tools/nested_trace_loop.py [reserved0]`: 300 calls of the one-call nested
quantize (quantize_4bit(compress_statistics=True)) on a 4096 x 4096 bf16 matrix; the per-kernel averages are the GPU time of the
4-bit encoder, the partial sums and the 8-bit encoder of the statistics. reserved0 = 9: the 8-bit tables rebuilt by every workgroup."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bitsandbytes_amd.functional as F
from bitsandbytes_amd.cextension import lib

knob = int(sys.argv[1]) if len(sys.argv) > 1 else 0
W = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
code8 = F._dynamic_map(W.device)
lib.bnb_mi355x_set_tuning(knob, 0, 0, 0)
for _ in range(300):
    torch.ops.bitsandbytes_amd.quantize_4bit_nested.default(W, code8, 64, "nf4", torch.uint8)
torch.cuda.synchronize()
lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
