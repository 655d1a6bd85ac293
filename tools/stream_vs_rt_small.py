#!/usr/bin/env python3
"""Where the streaming kernel hands over to the register-transposed MFMA kernel at 2 - 4 rows (c_api.hip::route_to_mfma): us per
launch of both on small and medium matrices, hipGraph-replayed over an HBM-resident rotation of layers."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bitsandbytes_amd as bnb  # noqa: E402
from stream_ab import make_layers, run  # noqa: E402


def main():
    print(torch.cuda.get_device_name(0), bnb.lib.bnb_mi355x_version().decode())
    shapes = [(1024, 4096), (4096, 1024), (2048, 2048), (1376, 4096), (512, 11008), (2048, 4096), (1024, 8192), (4096, 2048),
              (3072, 3072), (4096, 4096)]
    print(f"{'N x K':>14s} {'M weights':>9s} {'dq':>2s} {'M':>2s} {'stream':>8s} {'mfma rt':>8s}")
    for (N, K) in shapes:
        for dq in (False, True):
            layers = make_layers(N, K, 64, "nf4", dq)
            for M in (2, 3, 4):
                x = torch.randn(M, K, device="cuda").bfloat16()
                t = [min(run(layers, x, k) for _ in range(2)) for k in (3, 2)]
                print(f"{N:>7d}x{K:<6d} {N * K / 2**20:9.1f} {int(dq):>2d} {M:>2d} {t[0]:8.2f} {t[1]:8.2f}", flush=True)
            del layers


if __name__ == "__main__":
    main()
