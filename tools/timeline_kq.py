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


names = {0: "start", 1: "prologue DMA issued", 14: "chunk loop done", 15: "end"}
for _i in range(2):
    names[2 + 5 * _i] = f"chunk {2 + _i} top"
    names[3 + 5 * _i] = f"chunk {2 + _i} own DMA landed"
    names[4 + 5 * _i] = f"chunk {2 + _i} shares in registers (past barrier 2)"
    names[5 + 5 * _i] = f"chunk {2 + _i} computed"
ORDER = [0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 14, 15]

bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, a.cfg)
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
print(f"# kq kernel cfg={a.cfg} nested={a.nested} M={M} N={N} K={K}: {t.shape[0]} workgroups x {waves} wavefronts; s_memtime ticks relative to the "
      f"wavefront's own start")
print(f"{'stamp':44s} {'min':>7s} {'median':>7s} {'max':>7s}   median delta to previous stamp")
prev = None
for i in ORDER:
    ok = (t[:, :, i] > 0) & (t[:, :, 0] > 0)
    if ok.sum() == 0:
        continue
    rel = (t[:, :, i] - t[:, :, 0])[ok]
    line = f"{names[i]:44s} {rel.min().item():7.0f} {rel.median().item():7.0f} {rel.max().item():7.0f}"
    if prev is not None:
        both = ok & (t[:, :, prev] > 0)
        d = (t[:, :, i] - t[:, :, prev])
        line += f"   {d[both].median().item():8.0f}"
    print(line)
    prev = i
