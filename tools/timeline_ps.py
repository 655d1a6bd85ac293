#!/usr/bin/env python3
"""Per-wavefront timeline of the pre-scaled-operand MFMA kernel (gemm4_mfma_ps_kernel) from in-kernel s_memtime stamps
(profiling build only: make -C bitsandbytes_amd/csrc profiling; BNB_MI355X_LIBRARY=.../libbitsandbytes_mi355x_prof.so):
0 start, 1 weight stages requested, 2 table built, then per iteration i < 5: 3+2i past the barrier, 4+2i body done (iterations 0-2 fill
the in-wavefront pipeline, 3 and 4 are full ones); 13 stage loop done, 14 end. --ablate: profiling-only switches of the kernel (1 no activation DMA, 2 no weight DMA, 4 no decode, 8 no MFMA, 16 no
activation fragment reads).
    python tools/timeline_ps.py [--m 64] [--n 8192] [--k 8192] [--cfg 3000] [--ablate 0]"""
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
ap.add_argument("--ablate", type=str, default="0", help="comma-separated list of ablation masks, one timeline each")
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


names = {0: "start", 1: "weight stages requested", 15: "stage loop done", 2: "end"}
for _i in range(2):
    names[3 + 6 * _i] = f"iteration {8 + _i} top"
    names[4 + 6 * _i] = f"iteration {8 + _i} own DMA landed"
    names[5 + 6 * _i] = f"iteration {8 + _i} past barrier"
    names[6 + 6 * _i] = f"iteration {8 + _i} DMA issued"
    names[7 + 6 * _i] = f"iteration {8 + _i} half of the slots"
    names[8 + 6 * _i] = f"iteration {8 + _i} body done"
ORDER = [0, 1, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 2]


def timeline(ablate):
    bnb.lib.bnb_mi355x_set_tuning(0, 0, ablate, a.cfg)
    try:
        for i in range(L):
            step(i)
        torch.cuda.synchronize()
        buf.zero_()
        bnb.lib.bnb_mi355x_set_stamp_buffer(buf.data_ptr())
        step(0)                                              # ONE stamped launch (a later launch would overwrite the stamps)
        torch.cuda.synchronize()
    finally:
        bnb.lib.bnb_mi355x_set_stamp_buffer(None)
        bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
    t = buf.view(WG_MAX, waves, 16).cpu().double()
    used = (t[:, :, 0] > 0).any(dim=1)
    t = t[used]
    WG = t.shape[0]
    if WG == 0:
        print("no stamps: not a profiling build?")
        sys.exit(0)
    t0k = t[:, :, 0][t[:, :, 0] > 0].min()                   # first wavefront start of the launch
    print(f"# ps kernel cfg={a.cfg}, ablate={ablate}, M={a.m}, N={N}, K={K}: {WG} workgroups x {waves} wavefronts; s_memtime ticks "
          f"relative to the first wavefront start; launch span {(t[:, :, 2].max() - t0k).item():.0f}")
    print(f"{'stamp':34s} {'min':>7s} {'median':>7s} {'max':>7s}   median delta to previous stamp")
    prev = None
    for i in ORDER:
        ok = t[:, :, i] > 0
        if ok.sum() == 0:
            continue
        rel = (t[:, :, i] - t0k)[ok]
        line = f"{names[i]:34s} {rel.min().item():7.0f} {rel.median().item():7.0f} {rel.max().item():7.0f}"
        if prev is not None:
            both = ok & (t[:, :, prev] > 0)
            d = (t[:, :, i] - t[:, :, prev])
            line += f"   {d[both].median().item():8.0f}   weight loaders {d[:, :4][both[:, :4]].median().item():6.0f}  activation loaders {d[:, 4:][both[:, 4:]].median().item():6.0f}"
        print(line)
        prev = i


for ab in a.ablate.split(","):
    timeline(int(ab))
