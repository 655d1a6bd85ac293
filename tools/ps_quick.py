#!/usr/bin/env python3
"""Quick timing of the pre-scaled-operand MFMA kernel (tuning cfg 30) on a few shapes: launch-to-launch us over an HBM-resident
rotation of layers in a hipGraph. Used to compare builds (BNB_MI355X_LIBRARY=...)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bitsandbytes_amd as bnb  # noqa: E402
from stream_ab import make_layers, run  # noqa: E402

print(os.environ.get("BNB_MI355X_LIBRARY", "default library"))
for (N, K, Ms) in ((8192, 8192, (32, 64, 128, 256)), (4096, 4096, (64, 128)), (28672, 8192, (64, 128))):
    layers = make_layers(N, K, 64, "nf4", False, cap=24)
    for M in Ms:
        x = torch.randn(M, K, device="cuda").bfloat16()
        try:
            bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 3000)
            t = run(layers, x, 2)
        finally:
            bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
        print(f"{N:6d} x {K:5d} M = {M:4d}: {t:8.2f} us  {2.0 * M * N * K / t / 1e6:7.1f} TF/s", flush=True)
    del layers
