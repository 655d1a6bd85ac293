#!/usr/bin/env python3
"""Per-wavefront timeline of the streaming MFMA kernel (gemm4_mfma_sm_kernel) from in-kernel s_memtime stamps (measurement build
only: make -C bitsandbytes_amd/csrc profiling; BNB_MI355X_LIBRARY=.../libbitsandbytes_mi355x_prof.so).
    python tools/timeline_sm.py [--m 4] [--n 4096] [--k 4096]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from bitsandbytes_amd.backends import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=4)
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--k", type=int, default=4096)
a = ap.parse_args()
N, K = a.n, a.k
waves = 8 if a.m > 8 or N > 48 * 256 else 16
L = 32
g = torch.Generator(device="cuda").manual_seed(0)
layers = []
for _ in range(L):
    W = (torch.randn(N, K, device="cuda", generator=g) / K**0.5).bfloat16()
    layers.append(F.quantize_4bit(W, quant_type="nf4"))
    del W
x = torch.randn(a.m, K, device="cuda", generator=g).bfloat16()
WG = 256
buf = torch.zeros(WG * waves * 16, dtype=torch.int64, device="cuda")


def step(i):
    q, st = layers[i % L]
    return hip._gemm_4bit_fused(x, q, st.shape, st.absmax, st.blocksize, st.quant_type, None, None, None, None, kernel=2)


bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 5000)
for i in range(L):
    step(i)
torch.cuda.synchronize()
bnb.lib.bnb_mi355x_set_stamp_buffer(buf.data_ptr())
for i in range(L):
    step(i)
torch.cuda.synchronize()
bnb.lib.bnb_mi355x_set_stamp_buffer(None)
bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
t = buf.view(WG, waves, 16).cpu().double()
if (t[:, :, 0] > 0).sum() == 0:
    print("no stamps: not a profiling build?")
    sys.exit(0)
t0 = t[:, :, 0].min(dim=1, keepdim=True).values
names = ["start", "loads issued", "table written", "A landed (behind the barrier)", "frags read, next A requested", "past table barrier", "stage 0 landed",
         "item 0 done", "items done", "partials parked, past barrier", "end"]
print(f"# sm kernel M={a.m}, N={N}, K={K}: {waves} wavefronts per workgroup; s_memtime ticks relative to the first wavefront start of the SAME workgroup")
print(f"{'stamp':30s} {'min':>7s} {'median':>7s} {'p90':>7s} {'max':>7s}   median delta to previous stamp")
prev = None
for i in (0, 1, 2, 5, 3, 4, 6, 7, 8, 9, 10):
    ok = t[:, :, i] > 0
    if ok.sum() == 0:
        continue
    rel = (t[:, :, i] - t0)[ok]
    line = f"{names[i]:30s} {rel.min().item():7.0f} {rel.median().item():7.0f} {rel.quantile(0.9).item():7.0f} {rel.max().item():7.0f}"
    if prev is not None:
        both = ok & (t[:, :, prev] > 0)
        d = (t[:, :, i] - t[:, :, prev])[both]
        line += f"   {d.median().item():8.0f}"
    print(line)
    prev = i
allw = t[:, :, 0][t[:, :, 0] > 0]
print(f"# all wavefront starts span {allw.max().item() - allw.min().item():.0f} ticks")
