#!/usr/bin/env python3
"""Per-wavefront timeline of the pre-scaled-operand MFMA kernel (gemm4_mfma_ps_kernel) from in-kernel s_memtime stamps
(profiling build only: make -C bitsandbytes_amd/csrc profiling; BNB_MI355X_LIBRARY=.../libbitsandbytes_mi355x_prof.so):
0 start, 1 chunk 0 requested, 2 past the table barrier, 3 chunk 0's weights transposed, 4 K half 1's extra barrier, then per chunk
j < 2: 5+4j decode phase done, 6+4j past the barrier, 7+4j MFMA phase done, 8+4j past the barrier; 13 chunk loop done, 14 end.
    python tools/timeline_ps.py [--m 64] [--n 8192] [--k 8192] [--cfg 3000]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from bitsandbytes_amd.backends import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=64)
ap.add_argument("--n", type=int, default=8192)
ap.add_argument("--k", type=int, default=8192)
ap.add_argument("--cfg", type=int, default=3000)
a = ap.parse_args()
N, K = a.n, a.k
waves = 8
L = max(2, min(32, int(600e6 // (N * K // 2)) + 1))
g = torch.Generator(device="cuda").manual_seed(0)
layers = []
for _ in range(L):
    W = (torch.randn(N, K, device="cuda", generator=g) / K**0.5).bfloat16()
    layers.append(F.quantize_4bit(W, quant_type="nf4"))
    del W
x = torch.randn(a.m, K, device="cuda", generator=g).bfloat16()
WG_MAX = ((N + 127) // 128) * 64 * ((a.m + 31) // 32)  # upper bound on workgroups (any slice count)
buf = torch.zeros(WG_MAX * waves * 16, dtype=torch.int64, device="cuda")


def step(i):
    q, st = layers[i % L]
    return hip._gemm_4bit_fused(x, q, st.shape, st.absmax, st.blocksize, st.quant_type, None, None, None, None, kernel=2)


bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, a.cfg)
for i in range(L):
    step(i)
torch.cuda.synchronize()
bnb.lib.bnb_mi355x_set_stamp_buffer(buf.data_ptr())
for i in range(L):
    step(i)
torch.cuda.synchronize()
bnb.lib.bnb_mi355x_set_stamp_buffer(None)
bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
t = buf.view(WG_MAX, waves, 16).cpu().double()
used = (t[:, :, 0] > 0).any(dim=1)
t = t[used]
WG = t.shape[0]
if WG == 0:
    print("no stamps: not a profiling build?")
    sys.exit(0)
t0k = t[:, :, 0][t[:, :, 0] > 0].min()                   # first wavefront start of the whole launch
names = {0: "start", 1: "chunk 0 requested", 2: "past table barrier", 3: "chunk 0 transposed", 4: "phase offset barrier",
         5: "decode 0 done", 6: "past barrier", 7: "mfma 0 done", 8: "past barrier", 9: "decode 1 done", 10: "past barrier",
         11: "mfma 1 done", 12: "past barrier", 13: "chunk loop done", 14: "end"}
print(f"# ps kernel cfg={a.cfg}, M={a.m}, N={N}, K={K}: {WG} workgroups x {waves} wavefronts; s_memtime ticks "
      f"relative to the first wavefront start of the LAUNCH")
print(f"{'stamp':34s} {'min':>7s} {'median':>7s} {'p90':>7s} {'max':>7s}   median delta to previous stamp")
prev = None
for i in sorted(names):
    ok = t[:, :, i] > 0
    if ok.sum() == 0:
        continue
    rel = (t[:, :, i] - t0k)[ok]
    line = f"{names[i]:34s} {rel.min().item():7.0f} {rel.median().item():7.0f} {rel.quantile(0.9).item():7.0f} {rel.max().item():7.0f}"
    if prev is not None:
        both = ok & (t[:, :, prev] > 0)
        d = (t[:, :, i] - t[:, :, prev])[both]
        line += f"   {d.median().item():8.0f}"
    print(line)
    prev = i
