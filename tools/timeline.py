#!/usr/bin/env python3
"""Per-wavefront timeline of the dot kernel from in-kernel s_memtime stamps (profiling build path).
Stamps: 0 start, 1 table+x written, 2 after barrier, 3 all loads landed (forced vmcnt(0)), 4 compute done, 5 end."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from bitsandbytes_amd.backends import hip  # noqa: E402

N = K = 4096
L = 64
g = torch.Generator(device="cuda").manual_seed(0)
layers = []
for _ in range(L):
    W = (torch.randn(N, K, device="cuda", generator=g) / K**0.5).bfloat16()
    layers.append(F.quantize_4bit(W, quant_type="nf4"))
    del W
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--kernel", type=int, default=1)
ap.add_argument("--nt", type=int, default=0)
ap.add_argument("--cfgks", type=int, default=0)
args = ap.parse_args()
bnb.lib.bnb_mi355x_set_tuning(0, 0, args.nt, args.cfgks)
if args.kernel == 1:
    sys.exit("the stamped dot-kernel variant was retired (see profiles/r1_timeline_dotx_smemtime.txt for its record); "
             "use --kernel 2 (v2 MFMA) or tools/timeline_pc.py")
for M in (1, 8):
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    nw = N // 2 if args.kernel == 1 else 4096 * 4
    buf = torch.zeros(nw * 8, dtype=torch.int64, device="cuda")

    def step(i):
        q, st = layers[i % L]
        return hip._gemm_4bit_fused(x, q, st.shape, st.absmax, st.blocksize, st.quant_type, None, None, None, None, kernel=args.kernel)

    for i in range(L):
        step(i)
    torch.cuda.synchronize()
    bnb.lib.bnb_mi355x_set_stamp_buffer(buf.data_ptr())
    for i in range(L):
        step(i)  # the last launch's stamps remain
    torch.cuda.synchronize()
    bnb.lib.bnb_mi355x_set_stamp_buffer(None)
    t = buf.view(nw, 8).cpu().double()
    t = t[t[:, 0] > 0]
    nw = t.shape[0]
    if args.kernel == 2:
        names = ["start", "loads issued+lut", "after barrier", "loop done", "partials written", "after 2nd barrier", "end(wave0)"]
        d = t[:, 1:6] - t[:, 0:5]
        print(f"M={M} mfma: per-wave deltas median (ticks) over {nw} wavefronts: " + ", ".join(f"{names[i + 1]}: {d[:, i].median().item():.0f}" for i in range(5)))
        w0 = t[t[:, 6] > 0]
        print(f"   wave0 total start->end median {(w0[:, 6] - w0[:, 0]).median().item():.0f}, max {(w0[:, 6] - w0[:, 0]).max().item():.0f}; any-wave start->partials max {(t[:, 4] - t[:, 0]).max().item():.0f}")
        continue
    t0 = t[:, 0].min()
    rel = t[:, :6] - t0
    names = ["start", "lut+x written", "after barrier", "all loads landed", "compute done", "end"]
    print(f"M={M}: s_memtime ticks relative to first wave start (min / median / max over {nw} wavefronts)")
    for i, nme in enumerate(names):
        c = rel[:, i]
        print(f"  {nme:18s} {c.min().item():9.0f} {c.median().item():9.0f} {c.max().item():9.0f}")
    d = t[:, 1:6] - t[:, 0:5]
    print("  per-wave deltas (median): " + ", ".join(f"{names[i + 1]}: {d[:, i].median().item():.0f}" for i in range(5)))
    print(f"  total span: {(t[:, 5].max() - t0).item():.0f} ticks")
