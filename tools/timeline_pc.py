#!/usr/bin/env python3
"""Per-wavefront timeline of the producer/consumer MFMA kernel (gemm4_mfma_pc_kernel) from in-kernel
s_memtime stamps. Stamps per consumer wavefront: 0 start, 1 prologue issued + table written,
2+3k / 3+3k / 4+3k = chunk k: weights landed / past the barrier / compute done (k < 4), 14 loop done, 15 end.
    python tools/timeline_pc.py [--n 4096 --k 4096 --m 64 --cfgks 1108]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from bitsandbytes_amd.backends import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--k", type=int, default=4096)
ap.add_argument("--m", type=int, default=64)
ap.add_argument("--cfgks", type=int, default=1108)
a = ap.parse_args()
N, K, M = a.n, a.k, a.m
L = max(4, int(700e6 // (N * K // 2)))
g = torch.Generator(device="cuda").manual_seed(0)
layers = []
for _ in range(L):
    W = (torch.randn(N, K, device="cuda", generator=g) / K**0.5).bfloat16()
    layers.append(F.quantize_4bit(W, quant_type="nf4"))
    del W
bnb.lib.bnb_mi355x_set_tuning(0, 0, 1, a.cfgks)
x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
NW = 1 << 17
buf = torch.zeros(NW * 16, dtype=torch.int64, device="cuda")


def step(i):
    q, st = layers[i % L]
    return hip._gemm_4bit_fused(x, q, st.shape, st.absmax, st.blocksize, st.quant_type, None, None, None, None, kernel=2)


for i in range(L):
    step(i)
torch.cuda.synchronize()
bnb.lib.bnb_mi355x_set_stamp_buffer(buf.data_ptr())
for i in range(L):
    step(i)  # the last launch's stamps remain
torch.cuda.synchronize()
bnb.lib.bnb_mi355x_set_stamp_buffer(None)
t = buf.view(NW, 16).cpu().double()
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
names = ["start", "prologue+table"] + [f"c{k} {w}" for k in range(4) for w in ("W landed", "past barrier", "computed")] + ["loop done", "end"]
print(f"# M={M} N={N} K={K} cfgks={a.cfgks}: {t.shape[0]} consumer wavefronts; ticks of s_memtime (shader clock) relative to the first wavefront's start")
print(f"{'stamp':18s} {'min':>8s} {'median':>8s} {'max':>8s}   median delta to previous stamp")
prev = None
for i, nme in enumerate(names):
    c = t[:, i]
    ok = c > 0
    if ok.sum() == 0:
        continue
    rel = (c[ok] - t0)
    d = ""
    if prev is not None:
        both = ok & (t[:, prev] > 0)
        d = f"{(t[both, i] - t[both, prev]).median().item():8.0f}"
    print(f"{nme:18s} {rel.min().item():8.0f} {rel.median().item():8.0f} {rel.max().item():8.0f}   {d}")
    prev = i
