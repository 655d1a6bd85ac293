#!/usr/bin/env python3
"""Where does the fused route still win above 512 rows? (round 5) Matrices of <= 20 M weights (the FUSED_SMALL_WEIGHTS class of
backends/hip.py) and a few larger ones for contrast, M = 512 ... 1024: the fused call (64-row passes of the routed MFMA kernel) against
dequantize_4bit + the library GEMM (what the host dispatcher does above FUSED_MAX_M). us per call, hipGraph-replayed over an
HBM-resident rotation of distinct layers, round-robin, median of 3 regions >= 10 ms; plain and double-quantised statistics.
    python tools/tall_small_ab.py [--shapes 4096x2752,... --ms 8,16,32]
(second use, same columns: rows whose K is a multiple of the blocksize but not of 256 - the MFMA kernels do not serve them, the fused call
runs the streaming kernel in 4-row passes - to find the batch from which dequantize + GEMM is cheaper)"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bitsandbytes_amd.backends import hip  # noqa: E402
from stream_ab import make_layers  # noqa: E402

SHAPES = [(4096, 4096), (2048, 8192), (8192, 2048), (3072, 3072), (5120, 3584), (1376, 4096), (4096, 1376), (5120, 5120), (4096, 11008), (11008, 4096)]
MS = (512, 576, 640, 768, 896, 1024)


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


def timed(g, calls, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * calls) * 1e3


def main():
    global SHAPES, MS
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default=None, help="e.g. 4096x2752,11008x2752 (N x K)")
    ap.add_argument("--ms", default=None, help="e.g. 8,16,32")
    ap.add_argument("--bs", type=int, default=64, help="blocksize (32: plain statistics only are served by the MFMA route)")
    ap.add_argument("--plain-only", action="store_true")
    a = ap.parse_args()
    if a.shapes:
        SHAPES = [tuple(int(v) for v in t.split("x")) for t in a.shapes.split(",")]
    if a.ms:
        MS = tuple(int(v) for v in a.ms.split(","))
    print(torch.cuda.get_device_name(0))
    print(f"{'N x K (M weights)':>22s} {'nested':>6s} | " + " | ".join(f"M={m:<4d} fused unfus" for m in MS))
    for (N, K) in SHAPES:
        for nested in ((False,) if a.plain_only else (False, True)):
            layers = make_layers(N, K, a.bs, "nf4", nested, cap=8)
            cells = []
            for M in MS:
                x = torch.randn(M, K, device="cuda").bfloat16()
                out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)

                def args(st):
                    if st.nested:
                        return (st.shape, st.state2.absmax, st.blocksize, st.quant_type, None, st.absmax, st.state2.code, st.offset)
                    return (st.shape, st.absmax, st.blocksize, st.quant_type, None, None, None, None)

                def f_fused():
                    for q, st in layers:
                        hip._gemm_4bit_fused(x, q, *args(st), out=out)

                def f_unfused():
                    for q, st in layers:
                        hip._gemm_4bit_unfused(x, q, *args(st))

                gs = [capture(f_fused), capture(f_unfused)]
                t0 = timed(gs[0], len(layers), 3)
                reps = max(3, int(10000.0 / (t0 * len(layers))) + 1)
                samples = [[], []]
                for r in range(3):
                    for i in ((0, 1) if r % 2 == 0 else (1, 0)):
                        samples[i].append(timed(gs[i], len(layers), reps))
                t_f, t_u = (statistics.median(s) for s in samples)
                cells.append(f"{t_f:10.1f} {t_u:5.1f}" + ("*" if t_f < 0.97 * t_u else " "))
                del gs
            print(f"{N:>7d}x{K:<6d} ({N * K / 1e6:5.1f}) {int(nested):>6d} | " + " | ".join(cells), flush=True)
            del layers
    print("# * = the fused call is more than 3 % ahead")


if __name__ == "__main__":
    main()
