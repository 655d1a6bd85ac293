#!/usr/bin/env python3
"""Per-wavefront timeline of the streaming dot kernel (gemv4_stream_kernel) from in-kernel s_memtime stamps.
Needs the profiling build of the library:
    make -C bitsandbytes_amd/csrc profiling
    BNB_MI355X_LIBRARY=$PWD/bitsandbytes_amd/libbitsandbytes_mi355x_prof.so python tools/timeline_stream.py [--n 4096 --k 4096 --m 1]
Stamps per wavefront: 0 start, 1 ring prologue issued, 2 table written, 3 past the barrier, 4 activation slice in
registers, 5 first item decoded, 6 item loop done, 7 past the final barrier, 8 end."""
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
ap.add_argument("--m", type=int, default=1)
ap.add_argument("--tune", type=int, nargs=5, default=[0, 0, 0, -1, 0])
a = ap.parse_args()
N, K, M = a.n, a.k, a.m
L = max(4, int(700e6 // (N * K // 2)))
g = torch.Generator(device="cuda").manual_seed(0)
layers = []
for _ in range(L):
    W = (torch.randn(N, K, device="cuda", generator=g) / K**0.5).bfloat16()
    layers.append(F.quantize_4bit(W, quant_type="nf4"))
    del W
bnb.lib.bnb_mi355x_set_stream_tuning(*a.tune)
x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
NW = 1 << 16
buf = torch.zeros(NW * 16, dtype=torch.int64, device="cuda")


def step(i):
    q, st = layers[i % L]
    return hip._gemm_4bit_fused(x, q, st.shape, st.absmax, st.blocksize, st.quant_type, None, None, None, None, kernel=3)


for i in range(L):
    step(i)
torch.cuda.synchronize()
bnb.lib.bnb_mi355x_set_stamp_buffer(buf.data_ptr())
for i in range(L):
    step(i)  # the last launch's stamps remain
torch.cuda.synchronize()
bnb.lib.bnb_mi355x_set_stamp_buffer(None)
t = buf.view(NW, 16).cpu().double()
live = t[:, 0] > 0
# s_memtime counters are per XCD: normalise every wavefront to the earliest start on ITS XCD (workgroup b runs on XCD b % 8)
waves_per_wg = 16 if M == 1 and a.tune[4] != 8 else 8
rt = t[:, 13:15].clone()  # s_memrealtime (100 MHz, chip-wide) at wavefront start / end
wg = torch.arange(NW) // waves_per_wg
nwg = int(wg[live].max().item()) + 1
for b in range(nwg):  # s_memtime domains are not chip-wide: normalise every wavefront to ITS workgroup's first start
    sel = live & (wg == b)
    if sel.any():
        base = t[sel, 0].min()
        t[sel, :13] = torch.where(t[sel, :13] > 0, t[sel, :13] - base + 1, t[sel, :13])
        t[sel, 15] = torch.where(t[sel, 15] > 0, t[sel, 15] - base + 1, t[sel, 15])
rt = rt[live]
t = t[live]
r0 = rt[:, 0].min()
print(f"# realtime (10 ns ticks): wavefront starts span {(rt[:, 0].max() - r0).item() * 10:.0f} ns, last end at "
      f"{(rt[:, 1].max() - r0).item() * 10:.0f} ns after the first start; per-workgroup first start: median "
      f"{(rt[:, 0].view(-1, waves_per_wg).min(1).values - r0).median().item() * 10:.0f} ns")
t0 = 1.0
order = [0, 9, 10, 1, 2, 11, 3, 4, 15, 5, 6, 7, 8]
# CAUTION when reading waits off this table (round 5): stores count in vmcnt on gfx950, and every stamp is a global store -
# the counted x wait of the measurement build therefore waits for MORE than the product's (the stamps sit in the same in-order queue).
# Instruction-issue times are right; "own x landed" / "past barrier" are upper bounds.
names = {0: "start", 9: "before x DMA", 10: "x DMA issued", 11: "own x landed", 1: "ring issued",
         2: "table written", 3: "past barrier", 4: "x slice in regs", 15: "stage-0 weights landed", 5: "item 0 decoded", 6: "items done",
         7: "past final barrier", 8: "end"}
print(f"# M={M} N={N} K={K} tune={a.tune}: {t.shape[0]} wavefronts; s_memtime ticks relative to the first wavefront start of the SAME workgroup")
print(f"{'stamp':20s} {'min':>8s} {'median':>8s} {'p90':>8s} {'max':>8s}   median delta to previous stamp")
prev = None
for i in order:
    nme = names[i]
    c = t[:, i]
    ok = c > 0
    if ok.sum() == 0:
        continue
    rel = (c[ok] - t0)
    d = ""
    if prev is not None:
        both = ok & (t[:, prev] > 0)
        d = f"{(t[both, i] - t[both, prev]).median().item():8.0f}"
    print(f"{nme:20s} {rel.min().item():8.0f} {rel.median().item():8.0f} {rel.quantile(0.9).item():8.0f} {rel.max().item():8.0f}   {d}")
    prev = i
