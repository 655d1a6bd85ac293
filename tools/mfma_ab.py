#!/usr/bin/env python3
"""A/B of the MFMA kernels: the register-transposed kernel (cfg 20 / 21 / 22 = built-in / 8 / 16 wavefronts) against the
producer/consumer kernel (cfg 11 / 14) and the streaming dot kernel (kernel 3), per launch
over an HBM-resident rotation of distinct layers, hipGraph-replayed (launch-to-launch time in a dependent stream)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bitsandbytes_amd as bnb  # noqa: E402
from stream_ab import alg_bytes, make_layers, run  # noqa: E402


def timed(layers, x, kernel, knob1):
    try:
        bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, knob1)
        return run(layers, x, kernel)
    finally:
        bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--big", action="store_true", help="only the large-matrix / tall-batch cases")
    args = ap.parse_args()
    print(torch.cuda.get_device_name(0), bnb.lib.bnb_mi355x_version().decode())
    shapes = [(4096, 4096, False), (4096, 4096, True), (11008, 4096, False), (4096, 11008, False), (1376, 4096, False),
              (8192, 8192, False)]
    if args.quick:
        shapes = shapes[:1]
    if args.big:
        shapes = [(4096, 4096, False), (11008, 4096, False), (8192, 8192, False), (28672, 8192, False)]
    variants = [("auto", 0, 0), ("stream", 3, 0), ("v5 pc11", 2, 1100), ("v5 pc14", 2, 1400),
                ("rt", 2, 2000), ("rt 8w", 2, 2100), ("rt 16w", 2, 2200)]
    print(f"{'N x K':>14s} {'dq':>2s} {'M':>3s} " + " ".join(f"{n:>9s}" for n, _, _ in variants) + "   best GB/s (%HBM)")
    for (N, K, dq) in shapes:
        layers = make_layers(N, K, 64, "nf4", dq)
        for M in ((8, 16, 32, 64, 128) if args.big else (3, 4, 5, 8, 16, 32, 64)):
            x = torch.randn(M, K, device="cuda").bfloat16()
            row = []
            for name, kernel, knob1 in variants:
                if kernel == 3 and M > 8:
                    row.append(float("nan"))
                    continue
                if name.startswith("rt") and M > 64:
                    row.append(float("nan"))
                    continue
                row.append(timed(layers, x, kernel, knob1))
            best = min(v for v in row if v == v)
            gbs = alg_bytes(M, N, K, 64, dq) / best / 1e3
            print(f"{N:>7d}x{K:<6d} {int(dq):>2d} {M:>3d} " + " ".join(f"{v:9.2f}" for v in row) +
                  f"   {gbs:7.1f} ({gbs / 80:.1f})", flush=True)
        del layers


if __name__ == "__main__":
    main()
