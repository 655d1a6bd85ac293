#!/usr/bin/env python3
"""What each part of the K-quarter MFMA kernel costs PER CHUNK: compile-time ablated instances of the kernel (measurement build only:
bnb_mi355x_set_tuning knob0 = ablation bits; results are wrong by construction) timed at 4 and 16 chunks per workgroup (N = 8192,
4 K slices, K = 4096 / 16384); the slope is the time per 256-k chunk, the intercept the fixed cost (boundaries, start-up, epilogue, finalize).
    BNB_MI355X_LIBRARY=$PWD/bitsandbytes_amd/libbitsandbytes_mi355x_prof.so python tools/kq_ablate.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bitsandbytes_amd as bnb  # noqa: E402
from stream_ab import make_layers, run  # noqa: E402

BITS = {128: "VARIANT non-temporal weight DMA", 256: "VARIANT group 1 at priority 1", 512: "VARIANT priority 1 in the compute phase", 1024: "VARIANT priority 1 in the load phase", 2048: "VARIANT everything requested at start-up at once", 4096: "EXPERIMENT nested codes by dword requests (wrong codes)", 1: "no activation DMA", 2: "no weight/scale DMA", 4: "no look-ups", 8: "no MFMA", 16: "no LDS->register reads", 32: "no scale FMAs"}
SETS = [int(v) for v in os.environ.get('KQ_SETS', '0,1,2,3,4,8,12,16,32,19,44,63').split(',')]


def main():
    print(torch.cuda.get_device_name(0), os.environ.get("BNB_MI355X_LIBRARY", "default library (the knob is ignored!)"))
    M, N = 64, 8192
    res = {ab: [] for ab in SETS}
    for K in (4096, 16384):
        layers = make_layers(N, K, 64, "nf4", os.environ.get("KQ_NESTED", "0") == "1", cap=24)
        x = torch.randn(M, K, device="cuda").bfloat16()
        for ab in SETS:
            try:
                bnb.lib.bnb_mi355x_set_tuning(0, 0, ab, 4004)
                res[ab].append(min(run(layers, x, 2) for _ in range(3)))
            finally:
                bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
        del layers
    print(f"# M = {M}, N = {N}, 4 K slices: us at 4 / 16 chunks per workgroup, us per chunk, fixed us")
    for ab in SETS:
        t4, t16 = res[ab]
        slope = (t16 - t4) / 12
        what = " + ".join(v for k, v in BITS.items() if ab & k) or "everything on"
        print(f"  ablate {ab:3d}: {t4:7.2f} {t16:7.2f}   {slope:6.3f} per chunk   {t4 - 4 * slope:6.2f} fixed   {what}", flush=True)


if __name__ == "__main__":
    main()
