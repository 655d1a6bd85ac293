#!/usr/bin/env python3
"""Round 5, review item 1(d): the K-quarter MFMA kernel (tuning cfg 40) with a FORCED number of K slices against its built-in
plan and the shipped routing - fewer slices = fewer fp32 slab bytes written + read back by the finalize launch and fewer dirty
lines in front of its boundary, at the price of fewer workgroups. us per launch (kernel + finalize) over an HBM-resident rotation
of layers in a hipGraph, min of 3; `same` = the forced plan's result equals the built-in plan's bit for bit (slices are added in
slice order, so only the number of fp32 partial sums differs: not expected to be identical, reported for the record) and `err` its
relative error against fp32 dequantize + matmul.
    python tools/kq_ks_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from rt_variant_ab import one  # noqa: E402
from stream_ab import make_layers, run  # noqa: E402

CASES = [(8192, 8192, 64), (8192, 8192, 32), (4096, 4096, 64), (4096, 4096, 32), (11008, 4096, 64), (4096, 11008, 64)]


def main():
    print(torch.cuda.get_device_name(0), os.environ.get("BNB_MI355X_LIBRARY", "product library"))
    print(f"{'N x K':>14s} {'M':>3s} {'routing':>8s} {'kq plan':>8s} | forced K slices: " + " ".join(f"{k:>7d}" for k in (1, 2, 3, 4, 6, 8)))
    for (N, K, M) in CASES:
        layers = make_layers(N, K, 64, "nf4", False, cap=16)
        x = torch.randn(M, K, device="cuda").bfloat16()
        q, st = layers[0]
        ref = x.float() @ F.dequantize_4bit(q, st).float().t()
        row, errs = [], []
        for knob in (0, 4000, 4001, 4002, 4003, 4004, 4006, 4008):
            try:
                bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, knob)
                y = one(q, st, x).clone()
                errs.append(float((y.float() - ref).norm() / ref.norm()))
                row.append(min(run(layers, x, 2) for _ in range(3)))
            finally:
                bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
        bad = "" if max(errs) < 1e-2 else f"   <-- FAIL err {max(errs):.2e}"
        print(f"{N:>7d}x{K:<6d} {M:3d} {row[0]:8.2f} {row[1]:8.2f} | " + " " * 17 + " ".join(f"{t:7.2f}" for t in row[2:]) + bad, flush=True)
        del layers


if __name__ == "__main__":
    main()
