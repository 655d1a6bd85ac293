#!/usr/bin/env python3
"""Per-wavefront timeline of the fused backward kernel (gemm4_grad_input_kernel) from in-kernel s_memtime stamps (profiling
build only): 0 start, 1 first loads issued, 2 table built + barrier, then for the wavefront's first three 32-n blocks: grad_out /
scales staged in the private LDS patch (3, 6, 9), fragments read back (4, 7, 10), decode + MFMAs done (5, 8, 11); 12 block loop
done, 13 end.  python tools/timeline_bwd.py [--m 64]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=64)
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--k", type=int, default=4096)
a = ap.parse_args()
N, K = a.n, a.k
L = 32
g_ = torch.Generator(device="cuda").manual_seed(0)
layers = []
for _ in range(L):
    W = (torch.randn(N, K, device="cuda", generator=g_) / K**0.5).bfloat16()
    layers.append(F.quantize_4bit(W, quant_type="nf4"))
    del W
g = torch.randn(a.m, N, device="cuda", generator=g_).bfloat16()
WG = 1024
buf = torch.zeros(WG * 8 * 16, dtype=torch.int64, device="cuda")


def step(i):
    q, st = layers[i % L]
    return torch.ops.bitsandbytes_amd.gemm_4bit_grad_input.default(g, q, st.shape, st.absmax, 64, "nf4")


for i in range(L):
    step(i)
torch.cuda.synchronize()
bnb.lib.bnb_mi355x_set_stamp_buffer(buf.data_ptr())
for i in range(L):
    step(i)
torch.cuda.synchronize()
bnb.lib.bnb_mi355x_set_stamp_buffer(None)
t = buf.view(WG, 8, 16).cpu().double()
t = t[(t[:, :, 0] > 0).any(dim=1)]
if t.shape[0] == 0:
    print("no stamps: not a profiling build?")
    sys.exit(0)
t0 = torch.where(t[:, :, 0] > 0, t[:, :, 0], torch.full_like(t[:, :, 0], 1e30)).min(dim=1, keepdim=True).values
names = ["start", "loads issued", "table + barrier", "b0 staged", "b0 fragments", "b0 mfma done", "b1 staged",
         "b1 fragments", "b1 mfma done", "b2 staged", "b2 fragments", "b2 mfma done", "block loop done", "end"]
print(f"# backward kernel, M={a.m}, N={N}, K={K}: {t.shape[0]} workgroups x 8 wavefronts; s_memtime ticks relative to the first "
      f"wavefront start of the SAME workgroup")
print(f"{'stamp':20s} {'min':>7s} {'median':>7s} {'p90':>7s} {'max':>7s}   median delta to previous stamp")
prev = None
for i in range(14):
    ok = t[:, :, i] > 0
    if ok.sum() == 0:
        continue
    rel = (t[:, :, i] - t0)[ok]
    line = f"{names[i]:20s} {rel.min().item():7.0f} {rel.median().item():7.0f} {rel.quantile(0.9).item():7.0f} {rel.max().item():7.0f}"
    if prev is not None:
        both = ok & (t[:, :, prev] > 0)
        d = (t[:, :, i] - t[:, :, prev])[both]
        line += f"   {d.median().item():8.0f}"
    print(line)
    prev = i
