#!/usr/bin/env python3
"""Fixed cost vs per-chunk cost of the batched MFMA kernels: N = 8192 columns, 4 K slices (256 workgroups), K = 2048 ... 16384, i.e.
2 ... 16 chunks of 256 k per workgroup. Launch-to-launch us (kernel + finalize) over an HBM-resident rotation in a hipGraph; the
slope of the line is the time per chunk, the intercept everything else (launch boundary, start-up, epilogue, finalize launch).
    python tools/mfma_fixed_cost.py [--m 64]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bitsandbytes_amd as bnb  # noqa: E402
from stream_ab import make_layers, run  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", default="64,32")
ap.add_argument("--n", type=int, default=8192)
ap.add_argument("--cfgs", default="1104,4004")
a = ap.parse_args()
print(torch.cuda.get_device_name(0), os.environ.get("BNB_MI355X_LIBRARY", "default library"))
for M in (int(v) for v in a.m.split(",")):
    rows = {}
    for K in (2048, 4096, 8192, 16384):
        layers = make_layers(a.n, K, 64, "nf4", False, cap=24)
        x = torch.randn(M, K, device="cuda").bfloat16()
        for knob in (int(v) for v in a.cfgs.split(",")):
            try:
                bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, knob)
                rows.setdefault(knob, []).append(min(run(layers, x, 2) for _ in range(3)))
            finally:
                bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
        del layers
    for knob, ts in rows.items():
        ch = [2, 4, 8, 16]
        slope = (ts[3] - ts[1]) / (ch[3] - ch[1])
        print(f"M = {M:3d} N = {a.n} knob {knob}: " + "  ".join(f"{c:2d} chunks {t:7.2f} us" for c, t in zip(ch, ts))
              + f"   per chunk {slope:.3f} us, intercept {ts[1] - slope * ch[1]:.2f} us", flush=True)
