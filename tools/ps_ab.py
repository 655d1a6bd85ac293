#!/usr/bin/env python3
"""A/B of the pre-scaled-operand MFMA kernel (csrc/gemm4_mfma_ps.hip, tuning cfg 30 / 31) against the library's built-in
choice, the register-transposed kernel (cfg 20) and the producer/consumer kernel (cfg 11 / 14), and - for tall batches -
against dequantize + hipBLASLt (what the host dispatcher does above FUSED_MAX_M). Per launch over an HBM-resident rotation
of distinct layers, hipGraph-replayed (launch-to-launch time in a dependent stream)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bitsandbytes_amd as bnb  # noqa: E402
from bitsandbytes_amd.backends import hip  # noqa: E402
from stream_ab import alg_bytes, graph_time, make_layers, run  # noqa: E402


def timed(layers, x, kernel, knob1):
    try:
        bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, knob1)
        return run(layers, x, kernel)
    finally:
        bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)


def unfused(layers, x):
    def fn():
        for q, st in layers:
            if st.nested:
                hip._gemm_4bit_unfused(x, q, st.shape, st.state2.absmax, st.blocksize, st.quant_type, None, st.absmax,
                                       st.state2.code, st.offset)
            else:
                hip._gemm_4bit_unfused(x, q, st.shape, st.absmax, st.blocksize, st.quant_type, None, None, None, None)

    return graph_time(fn, len(layers))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--tall", action="store_true", help="M = 128 ... 2048 against dequantize + hipBLASLt")
    ap.add_argument("--slices", action="store_true", help="K-slice sweep of the ps kernel")
    args = ap.parse_args()
    print(torch.cuda.get_device_name(0), bnb.lib.bnb_mi355x_version().decode())
    shapes = [(4096, 4096, False), (8192, 8192, False), (11008, 4096, False), (4096, 11008, False), (4096, 4096, True),
              (1376, 4096, False), (28672, 8192, False)]
    if args.quick:
        shapes = shapes[:2]
    if args.tall:
        variants = [("auto", 0, 0), ("ps", 2, 3000), ("pc11", 2, 1100), ("unfused", -1, 0)]
        Ms = (128, 256, 512, 1024, 2048)
        shapes = [(4096, 4096, False), (8192, 8192, False), (11008, 4096, False), (4096, 11008, False), (28672, 8192, False)]
    elif args.slices:
        variants = [("ps", 2, 3000)] + [(f"ps ks{k}", 2, 3000 + k) for k in (1, 2, 3, 4, 6, 8, 16)]
        Ms = (16, 32, 64, 128)
    else:
        variants = [("auto", 0, 0), ("ps", 2, 3000), ("ps nopipe", 2, 3100), ("rt", 2, 2000), ("pc11", 2, 1100), ("pc14", 2, 1400)]
        Ms = (8, 16, 32, 64, 128)
    print(f"{'N x K':>14s} {'dq':>2s} {'M':>4s} " + " ".join(f"{n:>9s}" for n, _, _ in variants) + "   best GB/s (%HBM)  TF/s (%MFMA)")
    for (N, K, dq) in shapes:
        layers = make_layers(N, K, 64, "nf4", dq, cap=24 if args.tall else 64)
        for M in Ms:
            x = torch.randn(M, K, device="cuda").bfloat16()
            row = []
            for name, kernel, knob1 in variants:
                if name.startswith("rt") and M > 64:
                    row.append(float("nan"))
                    continue
                if kernel < 0:
                    row.append(unfused(layers, x))
                    continue
                if name == "auto" and M > hip.fused_max_m(N, K):
                    row.append(float("nan"))
                    continue
                row.append(timed(layers, x, kernel, knob1))
            best = min(v for v in row if v == v)
            gbs = alg_bytes(M, N, K, 64, dq) / best / 1e3
            tf = 2.0 * M * N * K / best / 1e6
            print(f"{N:>7d}x{K:<6d} {int(dq):>2d} {M:>4d} " + " ".join(f"{v:9.2f}" for v in row) +
                  f"   {gbs:7.1f} ({gbs / 80:.1f})  {tf:7.1f} ({tf / 25:.1f})", flush=True)
        del layers


if __name__ == "__main__":
    main()
