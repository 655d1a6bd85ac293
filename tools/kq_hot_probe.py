import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import bitsandbytes_amd as bnb
from stream_ab import make_layers, run
for (N, K, M) in ((8192, 8192, 64), (4096, 4096, 64), (2048, 8192, 64)):
    for cap in (24, 2, 1):
        layers = make_layers(N, K, 64, "nf4", False, cap=cap)[:cap]
        x = torch.randn(M, K, device="cuda").bfloat16()
        row = []
        for knob in (1100, 4000):
            bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, knob)
            row.append(min(run(layers, x, 2) for _ in range(3)))
            bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
        print(f"{N}x{K} M={M} layers in rotation {len(layers):2d}: pc {row[0]:7.2f}  kq {row[1]:7.2f}", flush=True)
        del layers
