#!/usr/bin/env python3
"""Per-wavefront timeline of the v3 MFMA kernel (gemm4_mfma_dma_kernel) from in-kernel s_memtime stamps:
0 start, 1 loads issued, 2 table written, 3 past the table barrier, 4 data landed, 5 MFMA loop done,
6 partials exchanged (barrier), 7 end (wavefront 0 only).   python tools/timeline_v3.py [--m 8]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from bitsandbytes_amd.backends import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=8)
a = ap.parse_args()
N = K = 4096
L = 64
g = torch.Generator(device="cuda").manual_seed(0)
layers = []
for _ in range(L):
    W = (torch.randn(N, K, device="cuda", generator=g) / K**0.5).bfloat16()
    layers.append(F.quantize_4bit(W, quant_type="nf4"))
    del W
x = torch.randn(a.m, K, device="cuda", generator=g).bfloat16()
NW = 256 * 16
buf = torch.zeros(NW * 16, dtype=torch.int64, device="cuda")


def step(i):
    q, st = layers[i % L]
    return hip._gemm_4bit_fused(x, q, st.shape, st.absmax, st.blocksize, st.quant_type, None, None, None, None, kernel=2)


for i in range(L):
    step(i)
torch.cuda.synchronize()
bnb.lib.bnb_mi355x_set_stamp_buffer(buf.data_ptr())
for i in range(L):
    step(i)
torch.cuda.synchronize()
bnb.lib.bnb_mi355x_set_stamp_buffer(None)
t = buf.view(NW, 16).cpu().double()
t = t[t[:, 0] > 0]
names = ["start", "loads issued", "table written", "past barrier", "data landed", "MFMA loop done", "partials exchanged", "end (wave 0)"]
print(f"# v3 kernel, M={a.m}, N=K=4096: {t.shape[0]} wavefronts; median per-wavefront delta to the previous stamp (shader cycles)")
prev = 0
for i in range(1, 8):
    ok = (t[:, i] > 0) & (t[:, prev] > 0)
    if ok.sum() == 0:
        continue
    d = t[ok, i] - t[ok, prev]
    print(f"{names[i]:20s} {d.median().item():8.0f}   (min {d.min().item():.0f}, max {d.max().item():.0f}, n={int(ok.sum())})")
    prev = i
tot = t[:, 6] - t[:, 0]
print(f"start -> partials exchanged: median {tot.median().item():.0f}, max {tot.max().item():.0f}")
