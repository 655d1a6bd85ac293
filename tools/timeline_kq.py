#!/usr/bin/env python3
"""Per-wavefront timeline of the K-quarter MFMA kernel (gemm4_mfma_kq_kernel) from in-kernel s_memtime stamps (measurement
build). Every stamp is taken relative to the SAME wavefront's first stamp (the counters of different XCDs are not synchronised).
    BNB_MI355X_LIBRARY=$PWD/bitsandbytes_amd/libbitsandbytes_mi355x_prof.so python tools/timeline_kq.py [--n 8192 --k 8192 --m 64 --cfg 4000]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from bitsandbytes_amd.backends import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=8192)
ap.add_argument("--k", type=int, default=8192)
ap.add_argument("--m", type=int, default=64)
ap.add_argument("--cfg", type=int, default=4000)
ap.add_argument("--nested", type=int, default=0)
ap.add_argument("--ablate", type=int, default=0, help="ablation bits of the stamped instance (3: no DMA in the loop, 16: no LDS -> register reads)")
a = ap.parse_args()
N, K, M = a.n, a.k, a.m
L = max(4, int(700e6 // (N * K // 2)))
g = torch.Generator(device="cuda").manual_seed(0)
layers = []
for _ in range(L):
    W = (torch.randn(N, K, device="cuda", generator=g) / K**0.5).bfloat16()
    layers.append(F.quantize_4bit(W, quant_type="nf4", compress_statistics=bool(a.nested)))
    del W
x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
waves, WG_MAX = 8, 1 << 13
buf = torch.zeros(WG_MAX * waves * 16, dtype=torch.int64, device="cuda")


def step(i):
    q, st = layers[i % L]
    if st.nested:
        return hip._gemm_4bit_fused(x, q, st.shape, st.state2.absmax, st.blocksize, st.quant_type, None, st.absmax, st.state2.code, st.offset, kernel=2)
    return hip._gemm_4bit_fused(x, q, st.shape, st.absmax, st.blocksize, st.quant_type, None, None, None, None, kernel=2)


names = {0: "start", 1: "start-up DMA issued", 2: "LAST half: load phase starts", 3: "share read, batch requested, first look-ups issued",
         9: "  reads issued", 
         4: "own requests landed (all but this phase's weights)", 5: "past the barrier: compute phase starts", 6: "computed", 7: "first barrier passed: half 0 in the LDS, table built", 13: "epilogue: partial sums exchanged through the LDS",
         8: "past the barrier", 14: "chunk loop done", 15: "end"}
ORDER = [0, 1, 7, 2, 3, 5, 6, 4, 8, 14, 15]

bnb.lib.bnb_mi355x_set_tuning(0, 0, 64 + a.ablate, a.cfg)  # knob0 = 64: the stamped instance of the kernel
try:
    for i in range(L):
        step(i)
    torch.cuda.synchronize()
    buf.zero_()
    bnb.lib.bnb_mi355x_set_stamp_buffer(buf.data_ptr())
    step(0)  # ONE stamped launch
    torch.cuda.synchronize()
finally:
    bnb.lib.bnb_mi355x_set_stamp_buffer(None)
    bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
t = buf.view(WG_MAX, waves, 16).cpu().double()
used = (t[:, :, 0] > 0).any(dim=1)
t = t[used]
if t.shape[0] == 0:
    print("no stamps: not a profiling build?")
    sys.exit(0)
print(f"# kq kernel cfg={a.cfg} ablate={a.ablate} nested={a.nested} M={M} N={N} K={K}: {t.shape[0]} workgroups x {waves} wavefronts; s_memtime ticks relative to the "
      f"wavefront's own start")
wg0 = torch.where(t[:, :, 0] > 0, t[:, :, 0], torch.full_like(t[:, :, 0], float("inf"))).min(dim=1, keepdim=True).values  # first wavefront start of the WORKGROUP (one XCD, one clock)
print(f"{'stamp':52s} {'median time since the workgroup started':>40s}   {'median delta to the previous stamp':>36s}")
print(f"{'':52s} {'all':>8s} {'group 0':>8s} {'group 1':>8s} {'min':>8s} {'max':>8s}   {'all':>8s} {'group 0':>8s} {'group 1':>8s}")
prev = None
for i in ORDER:
    ok = (t[:, :, i] > 0) & (t[:, :, 0] > 0)
    if ok.sum() == 0:
        continue
    rel = t[:, :, i] - wg0
    med = lambda x, m: x[m].median().item() if m.sum() else float("nan")  # noqa: E731
    line = (f"{names[i]:52s} {med(rel, ok):8.0f} {med(rel[:, :4], ok[:, :4]):8.0f} {med(rel[:, 4:], ok[:, 4:]):8.0f} "
            f"{rel[ok].min().item():8.0f} {rel[ok].max().item():8.0f}")
    if prev is not None:
        both = ok & (t[:, :, prev] > 0)
        d = (t[:, :, i] - t[:, :, prev])
        line += f"   {med(d, both):8.0f} {med(d[:, :4], both[:, :4]):8.0f} {med(d[:, 4:], both[:, 4:]):8.0f}"
    print(line)
    prev = i
ok0 = t[:, :, 0] > 0
for idx, what in ((13, 'of load-phase work'), (10, 'at the barrier ending the load phases'), (9, 'of compute-phase work'), (11, 'waiting for own DMA'), (12, 'at the barrier ending the compute phases')):
    v = t[:, :, idx]
    print(f'SUM over the slice, cycles {what:42s}: median {v[ok0].median().item():7.0f}  group 0 {v[:, :4][ok0[:, :4]].median().item():7.0f}  group 1 {v[:, 4:][ok0[:, 4:]].median().item():7.0f}')
# one workgroup in full: every wavefront's stamps, relative to the workgroup's start
w = t.shape[0] // 2
print(f"# workgroup {w}: rows = wavefronts 0..7 (group 0 = 0-3, group 1 = 4-7), columns = stamps {ORDER}")
for wv in range(waves):
    print("   " + " ".join(f"{(t[w, wv, i] - wg0[w, 0]).item():7.0f}" if t[w, wv, i] > 0 else "      -" for i in ORDER))
