#!/usr/bin/env python3
"""Round 6 A/B of the streaming MFMA kernel (csrc/gemm4_mfma_sm.hip) against the routing it replaces (knob0 bit 1: streaming
kernel at 2 rows / small matrices, register-transposed kernel above): us per launch over an HBM-resident rotation of distinct
layers, hipGraph-replayed, every configuration's graph captured once, timed regions >= 12 ms, round-robin, median.
Each output is also compared with fp32 dequantize + fp64 matmul (relative Frobenius error; bar 1e-2).
    python tools/sm_ab.py [--quick] [--rounds 5] [--m 2,4,8,16]"""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from bitsandbytes_amd.backends import hip  # noqa: E402
from stream_ab import alg_bytes, make_layers  # noqa: E402
from stream_prologue_ab import timed  # noqa: E402

K_NAMES = {1: "stream", 2: "generic", 3: "rt", 4: "pc", 6: "kq", 7: "sm"}


def one(q, st, x, kernel, out=None):
    if st.nested:
        return hip._gemm_4bit_fused(x, q, st.shape, st.state2.absmax, st.blocksize, st.quant_type, None, st.absmax,
                                    st.state2.code, st.offset, kernel=kernel, out=out)
    return hip._gemm_4bit_fused(x, q, st.shape, st.absmax, st.blocksize, st.quant_type, None, None, None, None, kernel=kernel, out=out)


def capture(layers, x, outs, kernel):
    def fn():
        for (q, st), o in zip(layers, outs):
            one(q, st, x, kernel, o)

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


def rel_err(q, st, x, y):
    W = F.dequantize_4bit(q, st).float()
    ref = x.double() @ W.double().t()
    return float((y.double() - ref).norm() / ref.norm())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--m", default="2,3,4,6,8,12,16")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--variants", default="", help="comma-separated knob0 values: A/B of experiment variants of the sm kernel instead of the routing table")
    ap.add_argument("--skip-odd", action="store_true")
    ap.add_argument("--shapes", default="", help="NxK[,NxK...]: these shapes (NF4, blocksize 64, plain statistics) instead of the built-in list")
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    print(torch.cuda.get_device_name(0), bnb.lib.bnb_mi355x_version().decode())
    cases = [(4096, 4096, 64, "nf4", False), (8192, 8192, 64, "nf4", False), (4096, 4096, 128, "fp4", True), (4096, 4096, 64, "nf4", True),
             (11008, 4096, 64, "nf4", False), (4096, 11008, 64, "nf4", False), (14336, 4096, 64, "nf4", False), (6144, 4096, 64, "nf4", False),
             (8192, 8192, 64, "nf4", True), (5120, 5120, 128, "nf4", False)]
    if args.quick:
        cases = cases[:3]
    if args.shapes:
        cases = [(int(v.split("x")[0]), int(v.split("x")[1]), 64, "nf4", False) for v in args.shapes.split(",")]
    ms = tuple(int(v) for v in args.m.split(","))
    print("# forced sm kernel on odd shapes: relative error against fp32 dequantize + fp64 matmul (bias included)")
    bad = 0
    for (N, K, bs, qt, dq, dtc) in [] if args.skip_odd else [(4100, 512, 64, "nf4", False, torch.bfloat16), (5000, 1024, 128, "fp4", True, torch.float16), (12345, 768, 64, "nf4", True, torch.bfloat16),
                                    (16, 256, 64, "nf4", False, torch.bfloat16), (4096, 4352, 256, "nf4", False, torch.float16), (20000, 256, 64, "fp4", False, torch.bfloat16),
                                    (70000, 512, 64, "nf4", False, torch.bfloat16), (4097, 8192, 512, "nf4", True, torch.bfloat16)]:
        W = (torch.randn(N, K, device="cuda") / K**0.5).to(dtc)
        q, st = F.quantize_4bit(W, blocksize=bs, quant_type=qt, compress_statistics=dq)
        Wd = F.dequantize_4bit(q, st).double()
        bias = torch.randn(N, device="cuda").to(dtc)
        for M in (1, 2, 3, 4, 5, 7, 8, 9, 13, 16, 17, 33):
            x = torch.randn(M, K, device="cuda").to(dtc)
            bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 5000)
            if st.nested:
                y = hip._gemm_4bit_fused(x, q, st.shape, st.state2.absmax, st.blocksize, st.quant_type, bias, st.absmax, st.state2.code, st.offset, kernel=2)
            else:
                y = hip._gemm_4bit_fused(x, q, st.shape, st.absmax, st.blocksize, st.quant_type, bias, None, None, None, kernel=2)
            fam = bnb.lib.bnb_mi355x_last_gemm_kernel()
            bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
            ref = x.double() @ Wd.t() + bias.double()
            err = float((y.double() - ref).norm() / ref.norm())
            rowerr = float(((y.double() - ref).norm(dim=1) / ref.norm(dim=1)).max())
            ok = err < 1e-2 and rowerr < 2e-2 and fam == 7
            bad += (not ok)
            if not ok:
                print(f"   {N}x{K} bs{bs} {qt} dq{int(dq)} {dtc} M={M}: err {err:.2e} worst row {rowerr:.2e} family {fam}   <-- FAIL", flush=True)
    print(f"   {bad} failures")
    # configurations: (label, kernel argument, knob0, knob1)
    configs = [("before", 0, 2, 0), ("sm", 2, 0, 5000), ("routed", 0, 0, 0)]
    if args.variants:
        configs = [("before", 0, 2, 0)] + [(f"sm v{v}", 2, int(v), 5000) for v in args.variants.split(",")]
    print(f"{'N x K':>14s} {'bs':>4s} {'qt':>3s} {'dq':>2s} {'M':>3s} " + " ".join(f"{c[0]:>12s}" for c in configs) + "   err before / sm    GB/s sm (%HBM)")
    for (N, K, bs, qt, dq) in cases:
        layers = make_layers(N, K, bs, qt, dq, dtype=dt)
        L = len(layers)
        for M in ms:
            x = torch.randn(M, K, device="cuda").to(dt)
            outs = [torch.empty(M, N, device="cuda", dtype=dt) for _ in layers]
            graphs, fams, errs = [], [], []
            for label, kernel, k0, k1 in configs:
                bnb.lib.bnb_mi355x_set_tuning(0, 0, k0, k1)
                y = one(*layers[0], x, kernel).clone()
                fams.append(K_NAMES.get(bnb.lib.bnb_mi355x_last_gemm_kernel(), "?"))
                errs.append(rel_err(*layers[0], x, y))
                graphs.append(capture(layers, x, outs, kernel))
            bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
            t0 = timed(graphs[0], L, 10)
            reps = max(10, int(12000.0 / (t0 * L)) + 1)
            samples = [[] for _ in configs]
            for r in range(args.rounds):
                order = list(range(len(configs)))
                if r % 2:
                    order.reverse()
                for i in order:
                    samples[i].append(timed(graphs[i], L, reps))
            med = [statistics.median(s) for s in samples]
            gbs = alg_bytes(M, N, K, bs, dq) / med[1] / 1e3
            cells = " ".join(f"{m:6.2f} {f:>5s}" for m, f in zip(med, fams))
            flag = "" if max(errs) < 1e-2 else "   <-- FAIL"
            print(f"{N:>7d}x{K:<6d} {bs:>4d} {qt:>3s} {int(dq):>2d} {M:>3d} {cells}   {errs[0]:.1e} / {errs[1]:.1e}   {gbs:7.1f} ({gbs / 80:.1f}){flag}", flush=True)
            del graphs
        del layers


if __name__ == "__main__":
    main()
