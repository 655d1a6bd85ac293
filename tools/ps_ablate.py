#!/usr/bin/env python3
"""What each operand stream costs the pre-scaled-operand MFMA kernel: the profiling build's ablation switches (set_tuning
knob0 bit 0: every lane fetches the first lane's scale, bit 1: ... activation piece, bit 2: ... weight piece - one memory
request per load instruction, results wrong) timed like tools/ps_ab.py.
    BNB_MI355X_LIBRARY=$PWD/bitsandbytes_amd/libbitsandbytes_mi355x_prof.so python tools/ps_ablate.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bitsandbytes_amd as bnb  # noqa: E402
from stream_ab import make_layers, run  # noqa: E402

print(torch.cuda.get_device_name(0), bnb.lib.bnb_mi355x_version().decode())
cases = [("full", 0), ("scales 1 req", 1), ("A 1 req", 2), ("W 1 req", 4), ("scales + A", 3), ("all three", 7)]
print(f"{'N x K':>14s} {'M':>4s} " + " ".join(f"{n:>13s}" for n, _ in cases) + "   (us per launch incl. finalize)")
for (N, K) in ((8192, 8192), (4096, 4096), (11008, 4096)):
    layers = make_layers(N, K, 64, "nf4", False)
    for M in (32, 64):
        x = torch.randn(M, K, device="cuda").bfloat16()
        row = []
        for _, k0 in cases:
            try:
                bnb.lib.bnb_mi355x_set_tuning(0, 0, k0, 3000)
                row.append(run(layers, x, 2))
            finally:
                bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
        print(f"{N:>7d}x{K:<6d} {M:>4d} " + " ".join(f"{v:13.2f}" for v in row), flush=True)
    del layers
