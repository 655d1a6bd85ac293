#!/usr/bin/env python3
"""Round 5 A/B of the streaming kernel (csrc/gemv4_stream.hip): wavefronts per workgroup (16 / 8) and ring depth. bf16, one activation row, NF4 bs 64, fp32 absmax (the sweep-only
instances); per-launch us over an HBM-resident rotation of distinct layers, hipGraph-replayed (launch-to-launch time in a dependent
stream).
    python tools/stream_prologue_ab.py [--quick] [--rounds 5]
Method: the first cut of this tool timed every configuration once, in a fixed order, over 5 replays (< 1 ms): the SAME kernel measured
twice in one row differed by 8 % (first-measured penalty). Now every configuration's graph is captured once, the timed region is
>= 15 ms of replays, and the configurations are measured round-robin over several rounds; the table shows the MEDIAN over the rounds
(min in brackets where it differs by more than 1 %).
First: every variant's output must equal the production instance's bit for bit on every shape (no variant moves arithmetic)."""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bitsandbytes_amd as bnb  # noqa: E402
from bitsandbytes_amd.backends import hip  # noqa: E402
from stream_ab import alg_bytes, make_layers  # noqa: E402

SHAPES = [(4096, 4096), (8192, 8192), (11008, 4096), (4096, 11008), (14336, 4096), (28672, 8192), (1376, 4096), (512, 11008)]
# (label, ring depth knob, nt knob, wavefronts[, 2048-k segments side by side]). (Runs 2 and 3 of profiles/r5_stream_prologue_ab.txt had
# more columns: the ring-late prologue - dropped, DESIGN 6b - and ring depths 3 / 4 at 16 wavefronts with it. Run 5: "16 sw1" = ONE segment
# column, i.e. a wavefront walks a whole row of K in K / 2048 phases and no other wavefront holds a partial sum of its rows.)
CONFIGS = [("built-in", 0, -1, 0), ("16 r2", 0, -1, 16), ("16 sw1", 0, -1, 16, 1), ("8 r4", 0, -1, 8)]


def tune(ns=0, nt=-1, waves=0, sw=0):
    bnb.lib.bnb_mi355x_set_stream_tuning(ns, sw, 0, nt, waves)


def one(q, st, x, out=None):
    return hip._gemm_4bit_fused(x, q, st.shape, st.absmax, st.blocksize, st.quant_type, None, None, None, None, kernel=3, out=out)


def capture(layers, x, outs):
    def fn():
        for (q, st), o in zip(layers, outs):
            one(q, st, x, o)

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


def timed(g, launches, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * launches) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--rounds", type=int, default=5)
    args = ap.parse_args()
    print(torch.cuda.get_device_name(0), bnb.lib.bnb_mi355x_version().decode(), os.environ.get("BNB_MI355X_LIBRARY", "product library"))
    shapes = SHAPES[:4] if args.quick else SHAPES
    print("# bit identity of every variant with the production instance of the same wavefront count")
    for (N, K) in shapes:
        layers = make_layers(N, K, 64, "nf4", False, cap=2)
        x = torch.randn(1, K, device="cuda").bfloat16()
        bad = []
        for waves in (16, 8):
            tune(waves=waves)
            ref = [one(q, st, x).clone() for q, st in layers]
            for label, ns, nt, w, *sw in CONFIGS:
                if w != waves:
                    continue
                tune(ns, nt, w, *sw)
                got = [one(q, st, x).clone() for q, st in layers]
                torch.cuda.synchronize()
                if not all(torch.equal(a, b) for a, b in zip(ref, got)):
                    bad.append(label)
        tune()
        print(f"   {N:6d} x {K:5d}: " + ("identical" if not bad else f"DIFFERENT: {bad}   <-- FAIL"), flush=True)
        del layers
    print(f"# us per launch, M = 1: median of {args.rounds} round-robin rounds, each >= 15 ms of graph replays [min where it differs by > 1 %]")
    print(f"{'N x K':>14s} " + " ".join(f"{c[0]:>11s}" for c in CONFIGS) + "   best: TB/s  %HBM")
    for (N, K) in shapes:
        layers = make_layers(N, K, 64, "nf4", False)
        L = len(layers)
        x = torch.randn(1, K, device="cuda").bfloat16()
        outs = [torch.empty(1, N, device="cuda", dtype=torch.bfloat16) for _ in layers]
        graphs = []
        for label, ns, nt, w, *sw in CONFIGS:  # the tuning is read at launch time, i.e. at capture: one graph per configuration
            tune(ns, nt, w, *sw)
            graphs.append(capture(layers, x, outs))
        tune()
        t0 = timed(graphs[0], L, 20)
        reps = max(20, int(15000.0 / (t0 * L)) + 1)
        samples = [[] for _ in CONFIGS]
        for r in range(args.rounds):
            order = list(range(len(CONFIGS)))
            if r % 2:
                order.reverse()
            for i in order:
                samples[i].append(timed(graphs[i], L, reps))
        med = [statistics.median(s) for s in samples]
        mn = [min(s) for s in samples]
        best = min(med)
        cells = [f"{m:6.2f}" + (f"[{lo:4.2f}]" if (m - lo) / m > 0.01 else "      ") for m, lo in zip(med, mn)]
        ab = alg_bytes(1, N, K, 64, False)
        print(f"{N:>7d}x{K:<6d} " + " ".join(f"{c:>11s}" for c in cells) + f"   {ab / best / 1e6:8.2f} {ab / best / 1e3 / 80:6.1f}", flush=True)
        del layers, graphs


if __name__ == "__main__":
    main()
