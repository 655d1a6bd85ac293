#!/usr/bin/env python3
"""K-quarter MFMA kernel (tuning cfg 40, csrc/gemm4_mfma_kq.hip) against the producer/consumer kernel (cfg 11) and the shipped
routing: correctness against a fp32 dequantize + matmul on the device, run-to-run bit identity, and launch-to-launch us over an
HBM-resident rotation of layers in a hipGraph.
    python tools/kq_ab.py [--quick]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from rt_variant_ab import one  # noqa: E402
from stream_ab import make_layers, run  # noqa: E402


def check(N, K, M, dq, qt="nf4", bs=64, dtype=torch.bfloat16, knob=4000):
    g = torch.Generator(device="cuda").manual_seed(N + K + M)
    W = (torch.randn(N, K, device="cuda", generator=g) / K**0.5).to(dtype)
    q, st = F.quantize_4bit(W, blocksize=bs, quant_type=qt, compress_statistics=dq)
    x = torch.randn(M, K, device="cuda", generator=g).to(dtype)
    ref = x.float() @ F.dequantize_4bit(q, st).float().t()
    try:
        bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, knob)
        y1 = one(q, st, x).clone()
        y2 = one(q, st, x).clone()
    finally:
        bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
    err = float((y1.float() - ref).norm() / ref.norm())
    return err, bool(torch.equal(y1, y2))


def main():
    quick = "--quick" in sys.argv
    print(torch.cuda.get_device_name(0), os.environ.get("BNB_MI355X_LIBRARY", "default library"))
    print("# correctness: relative error vs fp32 dequantize + matmul, bit identity of two runs")
    for (N, K, M, dq, qt, bs, dt) in ((256, 512, 64, False, "nf4", 64, torch.bfloat16), (384, 1024, 33, False, "nf4", 64, torch.bfloat16),
                                      (1000, 2816, 64, True, "nf4", 64, torch.bfloat16), (130, 512, 17, False, "fp4", 128, torch.float16),
                                      (512, 4096, 20, True, "fp4", 128, torch.float16), (8192, 8192, 64, False, "nf4", 64, torch.bfloat16),
                                      (4096, 11008 - 11008 % 256, 48, True, "nf4", 64, torch.bfloat16)):
        for knob in (4000, 4001, 4003):
            err, same = check(N, K, M, dq, qt, bs, dt, knob)
            print(f"  {N:5d} x {K:5d} M = {M:3d} {qt} bs {bs:3d} nested {int(dq)} {str(dt)[6:]:>8s} knob {knob}: err {err:.2e} same {same}"
                  + ("" if err < 1e-2 and same else "   <-- FAIL"), flush=True)
    print("# us per launch (kernel + finalize): shipped routing | cfg 20 (register-transposed) | cfg 11 (producer/consumer, 128 columns) | cfg 14 (64 columns) | cfg 40 (K-quarter)")
    cases = [(8192, 8192, (64, 32)), (4096, 4096, (64, 32)), (11008, 4096, (64,)), (4096, 11008, (64,))]
    if not quick:
        cases = [(8192, 8192, (17, 32, 40, 48, 64)), (28672, 8192, (17, 32, 64)), (11008, 4096, (17, 32, 48, 64)), (4096, 11008, (17, 32, 48, 64)),
                 (14336, 4096, (32, 64)), (4096, 14336, (32, 64)), (5120, 5120, (17, 32, 64)), (6144, 4096, (17, 32, 64)), (4096, 4096, (17, 32, 64)),
                 (8192, 2048, (32, 64)), (2048, 8192, (32, 64)), (1376, 4096, (32, 64)), (3072, 3072, (32, 64))]
    for (N, K, Ms) in cases:
        for dq in ((False, True) if quick else (False,)):
            layers = make_layers(N, K, 64, "nf4", dq, cap=24)
            for M in Ms:
                x = torch.randn(M, K, device="cuda").bfloat16()
                row = []
                for knob in (0, 2000, 1100, 1400, 4000):
                    try:
                        bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, knob)
                        row.append(min(run(layers, x, 2) for _ in range(2)))
                    finally:
                        bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
                print(f"  {N:6d} x {K:5d} M = {M:4d} nested {int(dq)}: " + " | ".join(f"{t:8.2f}" for t in row), flush=True)
            del layers


if __name__ == "__main__":
    main()
