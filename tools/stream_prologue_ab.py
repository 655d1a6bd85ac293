#!/usr/bin/env python3
"""Round 5 A/B of the streaming kernel's prologue (csrc/gemv4_stream.hip, kRingLate): the production instance against the same
instance with the table-building wavefronts' weight ring requested BEHIND the table build, so that the activation image is in the
CU's memory pipeline in front of all weight traffic. bf16, one activation row, NF4 bs 64, fp32 absmax (the sweep-only instances);
per-launch us over an HBM-resident rotation of distinct layers, hipGraph-replayed (launch-to-launch time in a dependent stream).
    python tools/stream_prologue_ab.py [--quick]
First: the variant's output must equal the production instance's bit for bit on every shape (the reorder moves no arithmetic)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bitsandbytes_amd as bnb  # noqa: E402
from bitsandbytes_amd.backends import hip  # noqa: E402
from stream_ab import alg_bytes, make_layers, run  # noqa: E402

SHAPES = [(4096, 4096), (8192, 8192), (11008, 4096), (4096, 11008), (14336, 4096), (28672, 8192), (1376, 4096), (512, 11008)]


def tune(ns=0, sw=0, rows=0, nt=-1, waves=0):
    bnb.lib.bnb_mi355x_set_stream_tuning(ns, sw, rows, nt, waves)


def one(q, st, x):
    return hip._gemm_4bit_fused(x, q, st.shape, st.absmax, st.blocksize, st.quant_type, None, None, None, None, kernel=3)


def main():
    quick = "--quick" in sys.argv
    print(torch.cuda.get_device_name(0), bnb.lib.bnb_mi355x_version().decode(), os.environ.get("BNB_MI355X_LIBRARY", "product library"))
    shapes = SHAPES[:4] if quick else SHAPES
    print("# bit identity: ring-late instance (nt = 2) vs production instance, 16 and 8 wavefronts")
    for (N, K) in shapes:
        layers = make_layers(N, K, 64, "nf4", False, cap=2)
        x = torch.randn(1, K, device="cuda").bfloat16()
        ok = True
        for waves in (16, 8):
            tune(waves=waves)
            ref = [one(q, st, x).clone() for q, st in layers]
            tune(nt=2, waves=waves)
            got = [one(q, st, x).clone() for q, st in layers]
            torch.cuda.synchronize()
            ok = ok and all(torch.equal(a, b) for a, b in zip(ref, got))
        tune()
        print(f"   {N:6d} x {K:5d}: {'identical' if ok else 'DIFFERENT   <-- FAIL'}", flush=True)
        del layers
    print("# us per launch (min of 3 graph timings), M = 1:   built-in choice | 16 wavefronts: production, ring-late | 8 wavefronts: production, ring-late")
    print(f"{'N x K':>14s} {'built-in':>9s} {'16 prod':>9s} {'16 late':>9s} {'8 prod':>9s} {'8 late':>9s}   best TB/s  %HBM")
    for (N, K) in shapes:
        layers = make_layers(N, K, 64, "nf4", False)
        x = torch.randn(1, K, device="cuda").bfloat16()
        row = []
        for kw in (dict(), dict(waves=16), dict(nt=2, waves=16), dict(waves=8), dict(nt=2, waves=8)):
            tune(**kw)
            row.append(min(run(layers, x, 3) for _ in range(3)))
        tune()
        best = min(row)
        print(f"{N:>7d}x{K:<6d} " + " ".join(f"{t:9.2f}" for t in row) + f"   {alg_bytes(1, N, K, 64, False) / best / 1e6:8.2f} {alg_bytes(1, N, K, 64, False) / best / 1e3 / 80:6.1f}", flush=True)
        del layers


if __name__ == "__main__":
    main()
