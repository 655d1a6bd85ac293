#!/usr/bin/env python3
"""Routing table of the batched (MFMA) kernels, round 4: per launch (kernel + finalize), hipGraph-replayed over an HBM-resident
rotation of distinct layers, us. Columns: the built-in routing | register-transposed (cfg 20) | producer/consumer (cfg 11) |
K-quarter (cfg 40) | [tall batches: dequantize + hipBLASLt].
    python tools/route_ab.py [small] [mid] [tall] [nested]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bitsandbytes_amd as bnb  # noqa: E402
from bitsandbytes_amd.backends import hip  # noqa: E402
from stream_ab import graph_time, make_layers, run  # noqa: E402


def unfused(layers, x):
    """dequantize_4bit + hipBLASLt: what the host dispatcher does above FUSED_MAX_M"""
    def fn():
        for q, st in layers:
            if st.nested:
                hip._gemm_4bit_unfused(x, q, st.shape, st.state2.absmax, st.blocksize, st.quant_type, None, st.absmax,
                                       st.state2.code, st.offset)
            else:
                hip._gemm_4bit_unfused(x, q, st.shape, st.absmax, st.blocksize, st.quant_type, None, None, None, None)

    return graph_time(fn, len(layers))


def timed(layers, x, knob, kernel=2):
    try:
        bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, knob)
        return min(run(layers, x, kernel) for _ in range(2))
    except Exception:  # noqa: BLE001
        return float("nan")
    finally:
        bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)


def table(title, cases, knobs, dqs=(False,), with_unfused=False):
    print(f"# {title}: " + " | ".join(n for n, _ in knobs) + (" | unfused" if with_unfused else ""), flush=True)
    for (N, K, Ms) in cases:
        for dq in dqs:
            layers = make_layers(N, K, 64, "nf4", dq, cap=24)
            for M in Ms:
                x = torch.randn(M, K, device="cuda").bfloat16()
                row = [timed(layers, x, k, 0 if k == 0 else 2) for _, k in knobs]
                if with_unfused:
                    row.append(unfused(layers, x))
                print(f"  {N:6d} x {K:5d} M = {M:4d} nested {int(dq)}: " + " | ".join(f"{t:8.2f}" for t in row), flush=True)
            del layers


def main():
    what = set(sys.argv[1:]) or {"small", "mid", "tall"}
    dqs = (False, True) if "nested" in what else (False,)
    print(torch.cuda.get_device_name(0), os.environ.get("BNB_MI355X_LIBRARY", "default library"))
    base = [("auto", 0), ("rt", 2000), ("pc", 1100), ("kq", 4000)]
    if "small" in what:
        table("33 ... 64 rows on small matrices", [(4096, 4096, (40, 48, 64)), (3072, 3072, (40, 48, 64)), (8192, 2048, (48, 64)),
                                                   (2048, 8192, (48, 64)), (1376, 4096, (48, 64)), (2048, 4096, (48, 64)), (5120, 5120, (48,))], base, dqs)
    if "mid" in what:
        table("5 ... 64 rows on large matrices", [(8192, 8192, (8, 16, 17, 32, 48, 64)), (28672, 8192, (8, 16, 32, 64)), (11008, 4096, (16, 32, 64)),
                                                  (4096, 11008, (16, 32, 64)), (6144, 4096, (32, 64))], base, dqs)
    if "tall" in what:
        table("tall batches", [(4096, 4096, (128, 256, 512)), (8192, 8192, (96, 128, 256, 512)), (11008, 4096, (128, 256, 512)),
                               (4096, 11008, (128, 256, 512)), (28672, 8192, (128, 256))],
              [("auto", 0), ("pc", 1100), ("kq", 4000)], dqs, with_unfused=True)


if __name__ == "__main__":
    main()
