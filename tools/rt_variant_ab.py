#!/usr/bin/env python3
"""A/B of two forms of the register-transposed MFMA kernel selected by bnb_mi355x_set_tuning knob0: 0 = shipped, 1 = round 2's
form (pointer loads, exec-masked rows, direct fragments up to 4 rows); other values were one-off experiment builds whose tables
are in profiles/r3_rt_*_ab.txt. Round 2's form only exists in the measurement build: run with
BNB_MI355X_LIBRARY=<repo>/bitsandbytes_amd/libbitsandbytes_mi355x_prof.so (the product library ignores the knob). Built-in routing otherwise: us per launch over an HBM-resident rotation of distinct layers, hipGraph-replayed
(launch-to-launch time in a dependent stream), plus a bit comparison of the two results (same arithmetic, same order of the
sums: they must be equal)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bitsandbytes_amd as bnb  # noqa: E402
from bitsandbytes_amd.backends import hip  # noqa: E402
from stream_ab import alg_bytes, make_layers, run  # noqa: E402


def one(q, st, x):
    if st.nested:
        return hip._gemm_4bit_fused(x, q, st.shape, st.state2.absmax, st.blocksize, st.quant_type, None, st.absmax,
                                    st.state2.code, st.offset, kernel=2)
    return hip._gemm_4bit_fused(x, q, st.shape, st.absmax, st.blocksize, st.quant_type, None, None, None, None, kernel=2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--m", default="3,4,5,8,16,32,64")
    ap.add_argument("--repeat", type=int, default=1, help="measure every cell this many times (alternating), print the minimum")
    ap.add_argument("--knobs", default="0,1", help="the two knob0 values to compare")
    args = ap.parse_args()
    print(torch.cuda.get_device_name(0), bnb.lib.bnb_mi355x_version().decode())
    cases = [(4096, 4096, 64, False), (4096, 4096, 64, True), (4096, 4096, 128, True), (8192, 8192, 64, False),
             (11008, 4096, 64, False), (4096, 11008, 64, False), (1376, 4096, 64, False), (8192, 8192, 64, True),
             (11008, 4096, 64, True), (4096, 11008, 64, True)]
    if args.quick:
        cases = cases[:4]
    ms = tuple(int(v) for v in args.m.split(","))
    knobs = tuple(int(v) for v in args.knobs.split(","))
    print(f"{'N x K':>14s} {'bs':>4s} {'dq':>2s} {'M':>3s} {'knob0 = ' + str(knobs[0]):>13s} {'knob0 = ' + str(knobs[1]):>9s} {'same bits':>10s}   GB/s (%HBM)")
    for (N, K, bs, dq) in cases:
        layers = make_layers(N, K, bs, "nf4", dq)
        for M in ms:
            if M > 16 and N * K > (20 << 20):
                continue  # (routed to the producer/consumer kernel)
            x = torch.randn(M, K, device="cuda").bfloat16()
            row, outs = [], []
            for _ in knobs:
                row.append(float("inf"))
                outs.append(None)
            for _ in range(args.repeat):
                for i, knob0 in enumerate(knobs):
                    try:
                        bnb.lib.bnb_mi355x_set_tuning(0, 0, knob0, 0)
                        row[i] = min(row[i], run(layers, x, 2))
                        outs[i] = one(*layers[0], x).clone()
                    finally:
                        bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
            gbs = alg_bytes(M, N, K, bs, dq) / min(row) / 1e3
            print(f"{N:>7d}x{K:<6d} {bs:>4d} {int(dq):>2d} {M:>3d} {row[0]:13.2f} {row[1]:9.2f} {str(torch.equal(outs[0], outs[1])):>10s}"
                  f"   {gbs:7.1f} ({gbs / 80:.1f})", flush=True)
        del layers


if __name__ == "__main__":
    main()
