#!/usr/bin/env python3
"""Per-wavefront timeline of the peer-chain form of the streaming kernel (gemv4_stream_kernel, kPeer instance) next to the plain
form, from in-kernel s_memtime stamps (measurement build):
    BNB_MI355X_LIBRARY=$PWD/bitsandbytes_amd/libbitsandbytes_mi355x_prof.so python tools/timeline_chain.py
One process (a gloo group of one rank): every layer is 4096 x 4096, consumes the previous exchange and produces the next.
Stamps: 0 start, 9 before x, 10 x requested, 1 ring issued, 2 table written, 3 past the barrier, 4 x slice in registers, 15 stage-0
weights landed, 5 item 0 decoded, 6 items done, 7 past the final barrier, 8 end."""
import os
import socket
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from bitsandbytes_amd.peer import PeerChain  # noqa: E402

N = K = 4096
with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
torch.cuda.set_device(0)
L = 80
g = torch.Generator(device="cuda").manual_seed(0)
layers = []
for _ in range(L):
    W = (torch.randn(N, K, device="cuda", generator=g) / K**0.5).bfloat16()
    layers.append(F.quantize_4bit(W, quant_type="nf4"))
    del W
x = torch.randn(K, device="cuda", generator=g).bfloat16()
chain = PeerChain(max_values=K)
NW = 1 << 16
buf = torch.zeros(NW * 16, dtype=torch.int64, device="cuda")


def run(mode):
    for i, (q, st) in enumerate(layers):
        if mode == "plain":
            bnb.matmul_4bit(x.view(1, -1), q, st)
        elif mode == "produce":
            chain.gemv(x, q, st, consume=False, produce=True)
        else:
            chain.gemv(x if i == 0 else None, q, st, consume=i > 0, produce=True, dtype=torch.bfloat16)
    if mode != "plain":
        chain.read(N, torch.bfloat16)


names = {0: "start", 9: "before x", 10: "x requested", 1: "ring issued", 2: "table written", 3: "past barrier", 4: "x slice in regs",
         15: "stage-0 weights landed", 5: "item 0 decoded", 6: "items done", 7: "past final barrier", 8: "end"}
order = [0, 9, 10, 1, 2, 3, 4, 15, 5, 6, 7, 8]
for mode in ("plain", "produce", "chain"):
    run(mode)
    torch.cuda.synchronize()
    buf.zero_()
    bnb.lib.bnb_mi355x_set_stamp_buffer(buf.data_ptr())
    run(mode)  # the last gemv launch's stamps remain (the read-out kernel has none)
    torch.cuda.synchronize()
    bnb.lib.bnb_mi355x_set_stamp_buffer(None)
    t = buf.view(NW, 16).cpu().double()
    live = t[:, 0] > 0
    wg = torch.arange(NW) // 16
    for b in range(int(wg[live].max().item()) + 1):
        sel = live & (wg == b)
        if sel.any():
            base = t[sel, 0].min()
            t[sel, :13] = torch.where(t[sel, :13] > 0, t[sel, :13] - base + 1, t[sel, :13])
            t[sel, 15] = torch.where(t[sel, 15] > 0, t[sel, 15] - base + 1, t[sel, 15])
    rt = t[live][:, 13:15]
    tt = t[live]
    r0 = rt[:, 0].min()
    print(f"## {mode}: {tt.shape[0]} wavefronts; realtime: starts span {(rt[:, 0].max() - r0).item() * 10:.0f} ns, last end {(rt[:, 1].max() - r0).item() * 10:.0f} ns after the first start")
    print(f"{'stamp':24s} {'min':>8s} {'median':>8s} {'p90':>8s} {'max':>8s} | wavefronts 0-7 median | 8-15 median")
    wave = (torch.arange(NW) % 16)[live]
    for i in order:
        c = tt[:, i]
        ok = c > 0
        if ok.sum() == 0:
            continue
        rel = c[ok] - 1.0
        lo, hi = ok & (wave < 8), ok & (wave >= 8)
        print(f"{names[i]:24s} {rel.min().item():8.0f} {rel.median().item():8.0f} {rel.quantile(0.9).item():8.0f} {rel.max().item():8.0f} | "
              f"{(tt[lo, i] - 1).median().item() if lo.any() else float('nan'):8.0f} | {(tt[hi, i] - 1).median().item() if hi.any() else float('nan'):8.0f}")
chain.close()
dist.destroy_process_group()
