#!/usr/bin/env python3
"""Per-launch time of the streaming dot kernel (kernel = 3) over an HBM-resident rotation of distinct layers,
hipGraph-replayed (launch-to-launch time in a dependent stream). (Round 1's dot kernel, its A/B partner until round 2,
is no longer in the library: profiles/r2_stream_ab.txt holds the last side-by-side run.)
Sections: shapes x M, tuning sweep at the headline shape, grouped launches vs separate launches."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from bitsandbytes_amd.backends import hip  # noqa: E402


def alg_bytes(M, N, K, bs, nested, elt=2):
    s = (N * K // bs) * (1 if nested else 4) + ((4 * ((N * K // bs + 255) // 256) + 1028) if nested else 0)
    return N * K // 2 + s + elt * M * K + elt * M * N


def make_layers(N, K, bs, qt, dq, dtype=torch.bfloat16, budget=600e6, cap=64):
    per = alg_bytes(1, N, K, bs, dq)
    L = max(2, min(cap, int(budget // per) + 1))
    g = torch.Generator(device="cuda").manual_seed(0)
    layers = []
    for _ in range(L):
        W = (torch.randn(N, K, device="cuda", generator=g) / K**0.5).to(dtype)
        layers.append(F.quantize_4bit(W, blocksize=bs, quant_type=qt, compress_statistics=dq))
        del W
    return layers


def graph_time(fn, launches, reps=5):
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
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * launches) * 1e3


def run(layers, x, kernel):
    outs = [torch.empty(x.shape[0], int(st.shape[0]), device="cuda", dtype=x.dtype) for _, st in layers]

    def fn():
        for (q, st), o in zip(layers, outs):
            if st.nested:
                hip._gemm_4bit_fused(x, q, st.shape, st.state2.absmax, st.blocksize, st.quant_type, None, st.absmax,
                                     st.state2.code, st.offset, kernel=kernel, out=o)
            else:
                hip._gemm_4bit_fused(x, q, st.shape, st.absmax, st.blocksize, st.quant_type, None, None, None, None,
                                     kernel=kernel, out=o)

    return graph_time(fn, len(layers))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    print(torch.cuda.get_device_name(0), bnb.lib.bnb_mi355x_version().decode())

    shapes = [(4096, 4096), (8192, 8192), (11008, 4096), (4096, 11008), (1376, 4096), (512, 11008), (28672, 8192)]
    if args.quick:
        shapes = shapes[:2]
    print(f"\n{'N x K':>14s} {'M':>2s} {'variant':>10s} {'us':>8s} {'GB/s':>9s} {'%HBM':>6s}")
    for (N, K) in shapes:
        for variant, qt, bs, dq in (("nf4-64", "nf4", 64, False), ("nf4-64-dq", "nf4", 64, True)):
            if variant != "nf4-64" and (N, K) not in ((4096, 4096), (11008, 4096)):
                continue
            layers = make_layers(N, K, bs, qt, dq)
            for M in (1, 2, 4):
                x = torch.randn(M, K, device="cuda").bfloat16()
                t_new = run(layers, x, 3)
                gbs = alg_bytes(M, N, K, bs, dq) / t_new / 1e3
                print(f"{N:>7d}x{K:<6d} {M:2d} {variant:>10s} {t_new:8.2f} {gbs:9.1f} {gbs / 80:6.1f}", flush=True)
            del layers

    print("\n-- other activation dtypes, 4096 x 4096, M = 1")
    for dt in (torch.float16, torch.float32):
        layers = make_layers(4096, 4096, 64, "nf4", False, dtype=dt)
        x = torch.randn(1, 4096, device="cuda").to(dt)
        print(f"   {str(dt):16s} {run(layers, x, 3):7.2f} us", flush=True)
        del layers

    print("\n-- tuning sweep (bf16, M = 1): ring depth / segments side by side / rows per workgroup / nt / wavefronts")
    for (N, K) in ((4096, 4096), (8192, 8192), (11008, 4096)):
        layers = make_layers(N, K, 64, "nf4", False)
        x = torch.randn(1, K, device="cuda").bfloat16()
        for tune in [(0, 0, 0, -1, 0), (2, 0, 0, -1, 0), (3, 0, 0, -1, 0), (6, 0, 0, -1, 0), (0, 0, 0, 0, 0), (0, 0, 0, -1, 8),
                     (0, 0, 0, 0, 8), (0, 1, 0, -1, 0), (0, 0, (N + 511) // 512, -1, 0), (0, 0, (N + 127) // 128, -1, 0)]:
            bnb.lib.bnb_mi355x_set_stream_tuning(*tune)
            t = run(layers, x, 3)
            print(f"   {N}x{K} ns={tune[0]} sw={tune[1]} rows={tune[2]} nt={tune[3]} waves={tune[4]}: {t:7.2f} us "
                  f"{alg_bytes(1, N, K, 64, False) / t / 1e3:8.1f} GB/s", flush=True)
        bnb.lib.bnb_mi355x_set_stream_tuning(0, 0, 0, -1, 0)
        del layers

    print("\n-- grouped launch vs separate launches (us per GROUP; bf16, M = 1)")
    for name, K, Ns in (("QKV 4096+1024+1024 (Llama-3-8B)", 4096, (4096, 1024, 1024)), ("gate/up 2 x 14336", 4096, (14336, 14336)),
                        ("3 x 4096 square", 4096, (4096, 4096, 4096))):
        per = sum(alg_bytes(1, n, K, 64, False) for n in Ns)
        L = max(2, min(32, int(600e6 // per) + 1))
        g = torch.Generator(device="cuda").manual_seed(1)
        groups = []
        for _ in range(L):
            grp = []
            for n in Ns:
                W = (torch.randn(n, K, device="cuda", generator=g) / K**0.5).bfloat16()
                grp.append(F.quantize_4bit(W, blocksize=64, quant_type="nf4"))
                del W
            groups.append(grp)
        x = torch.randn(1, K, device="cuda").bfloat16()

        def sep():
            for grp in groups:
                for q, st in grp:
                    bnb.matmul_4bit(x, q, st)

        def grpd():
            for grp in groups:
                bnb.matmul_4bit_grouped(x, [q for q, _ in grp], [st for _, st in grp])

        t_sep, t_grp = graph_time(sep, L), graph_time(grpd, L)
        print(f"   {name:34s} separate {t_sep:7.2f} us ({per / t_sep / 1e3:7.1f} GB/s)   grouped {t_grp:7.2f} us "
              f"({per / t_grp / 1e3:7.1f} GB/s)", flush=True)
        del groups

    print("\n-- headline sweep through the production dispatch (4096 x 4096 NF4 bf16)")
    layers = make_layers(4096, 4096, 64, "nf4", False)
    for M in (1, 2, 3, 4, 5, 8, 16, 32, 64):
        x = torch.randn(M, 4096, device="cuda").bfloat16()
        t = run(layers, x, 0)
        print(f"   M={M:3d} {t:7.2f} us {alg_bytes(M, 4096, 4096, 64, False) / t / 1e3:8.1f} GB/s {2 * M * 4096 * 4096 / t / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
