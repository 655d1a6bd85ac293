#!/usr/bin/env python3
"""Experiment queued for round 4: the producer/consumer MFMA kernel with the second consumer wavefront of every
SIMD delayed by d x 128 cycles behind the first after every chunk barrier (measurement build only: bnb_mi355x_set_tuning knob0 =
16 d), so that the two wavefronts of a SIMD stop being in the same phase (look-ups / MFMAs / scale FMAs) at the same time.
us per launch pair (kernel + finalize), hipGraph-replayed over an HBM-resident rotation of layers; results must be bit-identical.
    BNB_MI355X_LIBRARY=$PWD/bitsandbytes_amd/libbitsandbytes_mi355x_prof.so python tools/pc_phase_shift_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bitsandbytes_amd as bnb  # noqa: E402
from rt_variant_ab import one  # noqa: E402
from stream_ab import make_layers, run  # noqa: E402


def main():
    print(torch.cuda.get_device_name(0), bnb.lib.bnb_mi355x_version().decode())
    if "prof" not in os.environ.get("BNB_MI355X_LIBRARY", ""):
        print("note: the delay only exists in the measurement build (BNB_MI355X_LIBRARY=.../libbitsandbytes_mi355x_prof.so); "
              "the product library ignores the knob")
    cases = [(8192, 8192, 64), (8192, 8192, 32), (11008, 4096, 64), (4096, 11008, 64), (28672, 8192, 64)]
    delays = (0, 1, 2, 3, 4, 6, 8, 12)
    print(f"{'N x K':>14s} {'M':>4s} " + " ".join(f"{'d=' + str(d):>8s}" for d in delays) + "  same bits")
    for (N, K, M) in cases:
        layers = make_layers(N, K, 64, "nf4", False)
        x = torch.randn(M, K, device="cuda").bfloat16()
        row, outs = [], []
        for d in delays:
            try:
                bnb.lib.bnb_mi355x_set_tuning(0, 0, 16 * d, 1100)  # cfg 11: 8 consumers x 16 columns, built-in K slices
                row.append(min(run(layers, x, 2) for _ in range(2)))
                outs.append(one(*layers[0], x).clone())
            finally:
                bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        print(f"{N:>7d}x{K:<6d} {M:>4d} " + " ".join(f"{v:8.2f}" for v in row) + f"  {same}", flush=True)
        del layers


if __name__ == "__main__":
    main()
