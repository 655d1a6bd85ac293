#!/usr/bin/env python3
"""Round 6: groups of weight matrices that share x (Q/K/V/O, gate/up) at 1 ... 64 rows, us per GROUP, hipGraph-replayed over an HBM-resident
rotation of distinct groups, regions >= 12 ms, round-robin, median. Three columns: the grouped call under round 5's routing (one launch of the
streaming kernel where no member went to an MFMA kernel, else matrix by matrix), the members one by one through the single-matrix op, and the
shipped grouped call (bnb_mi355x_gemm_4bit_grouped: one launch of the streaming MFMA kernel from two rows on where the library's rule says so;
--force-sm: that launch forced, for measuring beyond the rule). The basis of c_api.hip: grouped_sm_passes and of the grouped route.
    python tools/grouped_ab.py [--rounds 5] [--m 1,2,4,8,16,32] [--force-sm]"""
import argparse
import ctypes as ct
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from bitsandbytes_amd.backends import hip  # noqa: E402
from stream_prologue_ab import timed  # noqa: E402


class NeverMfma:
    """hip.lib with the route query answering 'streaming kernel': forces the grouped launch."""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        if name == "bnb_mi355x_gemm_4bit_route":
            return lambda *a: 0
        return getattr(self._lib, name)


def capture(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    return g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--m", default="1,2,3,4")
    ap.add_argument("--force-sm", action="store_true", help="third column: the grouped call with the streaming MFMA kernel forced (knob cfg 50)")
    args = ap.parse_args()
    print(torch.cuda.get_device_name(0), bnb.lib.bnb_mi355x_version().decode())
    groups = [("Q/K/V/O 4 x 4096^2", [(4096, 4096)] * 4), ("GQA Q + K + V 4096 + 2 x 1024", [(4096, 4096), (1024, 4096), (1024, 4096)]),
              ("gate/up 2 x 11008 x 4096", [(11008, 4096)] * 2), ("gate/up shard 2 x 1376 x 4096", [(1376, 4096)] * 2),
              ("Q/K/V shard 3 x 512 x 4096", [(512, 4096)] * 3), ("gate/up 2 x 14336 x 4096", [(14336, 4096)] * 2),
              ("Q/K/V 3 x 8192^2", [(8192, 8192)] * 3)]
    real_lib = hip.lib
    print(f"{'group':>34s} {'M':>2s} {'round 5':>9s} {'separate':>9s} {'shipped':>9s}   us per group. round 5 = the grouped call under that round's routing (knob0 bit 1: one launch of\n"
          f"{'':>68s}# the streaming kernel where no member went to an MFMA kernel, else matrix by matrix); separate = the single-matrix op per member; shipped = the grouped call")
    for label, shapes in groups:
        K = shapes[0][1]
        per = sum(n * K // 2 + n * K // 16 for n, _ in shapes)
        L = max(2, min(24, int(400e6 // per) + 1))
        gen = torch.Generator(device="cuda").manual_seed(0)
        sets = []
        for _ in range(L):
            ws, sts = [], []
            for (N, _) in shapes:
                W = (torch.randn(N, K, device="cuda", generator=gen) / K**0.5).bfloat16()
                q, st = F.quantize_4bit(W, blocksize=64, quant_type="nf4")
                ws.append(q)
                sts.append(st)
                del W
            sets.append((ws, sts))
        for M in (int(v) for v in args.m.split(",")):
            x = torch.randn(M, K, device="cuda").bfloat16()
            outs = [[torch.empty(M, n, device="cuda", dtype=torch.bfloat16) for n, _ in shapes] for _ in sets]

            def grouped():
                for (ws, sts), o in zip(sets, outs):
                    bnb.matmul_4bit_grouped(x, ws, sts, None, outs=o)

            def separate():
                for (ws, sts), o in zip(sets, outs):
                    for w, s, oo in zip(ws, sts, o):
                        hip._gemm_4bit_fused(x, w, s.shape, s.absmax, s.blocksize, s.quant_type, None, None, None, None, kernel=0, out=oo)

            # the routing of round 5 (knob0 bit 1: no streaming MFMA kernel) + the route query answering "streaming kernel": the
            # grouped launch wherever the library's own rule (bit-identity with the members' single-matrix route) allows it
            real_lib.bnb_mi355x_set_tuning(0, 0, 2, 0)
            hip.lib = NeverMfma(real_lib)
            g_grouped = capture(grouped)
            hip.lib = real_lib
            real_lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
            g_sep = capture(separate)
            if args.force_sm:  # (experiment: the grouped streaming MFMA launch beyond its routed range - row passes of 16 over grid.y)
                real_lib.bnb_mi355x_set_tuning(0, 0, 0, 5000)
            g_now = capture(grouped)  # the shipped rule
            real_lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
            ns = (ct.c_int * len(shapes))(*[n for n, _ in shapes])
            route = real_lib.bnb_mi355x_gemm_4bit_grouped_route(2, len(shapes), ns, M, K, 64)
            graphs = (g_grouped, g_sep, g_now)
            t0 = timed(g_grouped, L, 5)
            reps = max(5, int(12000.0 / (t0 * L)) + 1)
            samples = ([], [], [])
            for r in range(args.rounds):
                order = (0, 1, 2) if r % 2 == 0 else (2, 1, 0)
                for i in order:
                    samples[i].append(timed(graphs[i], L, reps))
            a, b, c = (statistics.median(s) for s in samples)
            print(f"{label:>34s} {M:>2d} {a:9.2f} {b:9.2f} {c:9.2f}   {('matrix by matrix', 'one streaming launch', 'one streaming MFMA launch')[route]}", flush=True)
            del g_grouped, g_sep, g_now, graphs
        del sets


if __name__ == "__main__":
    main()
