#!/usr/bin/env python3
"""Run-to-run determinism of every kernel of the path under a perturbed machine: each call is repeated RUNS times while a second stream
keeps the memory system busy with large copies (other arrival orders of wavefronts and workgroups, other cache states), and every
result must equal the first one bit for bit. Covers the routed kernel for M = 1 ... 512 on three shapes (streaming, register-transposed,
producer/consumer and K-quarter families in the ranges they ship in; plain and double-quantised statistics), the fused backward, and the
standalone quantize / dequantize kernels (incl. the one-call nested quantize, whose offset is summed in a fixed order).
    python tests/checks/determinism_stress.py [--runs 200]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from bitsandbytes_amd.backends import hip  # noqa: E402

DEV = "cuda"
FAMILY = {1: "stream", 2: "generic", 3: "rt", 4: "pc", 6: "kq", 7: "sm", 8: "tall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=200)
    a = ap.parse_args()
    print(torch.cuda.get_device_name(0), flush=True)
    noise_stream = torch.cuda.Stream()
    src = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    dst = torch.empty_like(src)
    bad_total = 0

    def repeat(fn, label):
        nonlocal bad_total
        first = [t.clone() for t in fn()]
        differing = 0
        for i in range(a.runs):
            if i % 8 == 0:
                with torch.cuda.stream(noise_stream):
                    dst.copy_(src, non_blocking=True)
            got = fn()
            differing += int(not all(torch.equal(g.view(torch.uint8), f.view(torch.uint8)) for g, f in zip(got, first)))
        torch.cuda.synchronize()
        bad_total += differing
        if differing:
            print(f"{label}: {differing} of {a.runs} runs differ   <-- FAIL", flush=True)
        return differing

    seen = {}
    for (N, K) in ((4096, 4096), (11008, 4096), (8192, 8192)):
        for dq in (False, True):
            torch.manual_seed(N + int(dq))
            W = (torch.randn(N, K, device=DEV) / K**0.5).bfloat16()
            q, st = F.quantize_4bit(W, blocksize=64, quant_type="nf4", compress_statistics=dq)
            if dq:
                args = (st.shape, st.state2.absmax, 64, "nf4", None, st.absmax, st.state2.code, st.offset)
            else:
                args = (st.shape, st.absmax, 64, "nf4", None, None, None, None)
            for M in (1, 2, 3, 4, 5, 8, 16, 17, 32, 33, 48, 64, 65, 128, 512):
                if M > 64 and N * K > 64 << 20:
                    continue
                x = torch.randn(M, K, device=DEV).bfloat16()
                d = repeat(lambda: [hip._gemm_4bit_fused(x, q, *args)], f"forward {N} x {K} M = {M} nested {int(dq)}")
                fam = FAMILY.get(bnb.lib.bnb_mi355x_last_gemm_kernel(), "?")
                c = seen.setdefault(fam, [0, 0])
                c[0] += 1
                c[1] += d
                if M in (1, 8, 64, 128) and N == 4096:
                    g = torch.randn(M, N, device=DEV).bfloat16()
                    op = torch.ops.bitsandbytes_amd.gemm_4bit_grad_input.default
                    if dq:
                        d = repeat(lambda: [op(g, q, st.shape, st.state2.absmax, 64, "nf4", absmax_8bit=st.absmax, absmax_code=st.state2.code,
                                               absmax_offset=st.offset)], f"backward M = {M} nested")
                    else:
                        d = repeat(lambda: [op(g, q, st.shape, st.absmax, 64, "nf4")], f"backward M = {M}")
                    c = seen.setdefault("grad_input", [0, 0])
                    c[0] += 1
                    c[1] += d
            del W, q, st
    # forced geometries of the MFMA families (tuning knob cfg * 100 + K slices: 11-14 producer/consumer, 20-22 register-transposed,
    # 40 K-quarter), incl. cross-workgroup K slices (fp32 slabs + the finalize launch) on shapes they would not be routed to
    for (M, N, K) in ((16, 512, 4096), (33, 384, 1024), (64, 1000, 2816), (64, 4096, 4096), (100, 1376, 4096)):
        torch.manual_seed(M + N)
        W = (torch.randn(N, K, device=DEV) / K**0.5).bfloat16()
        x = torch.randn(M, K, device=DEV).bfloat16()
        for dq in (False, True):
            q, st = F.quantize_4bit(W, blocksize=64, quant_type="nf4", compress_statistics=dq)
            if dq:
                args = (st.shape, st.state2.absmax, 64, "nf4", None, st.absmax, st.state2.code, st.offset)
            else:
                args = (st.shape, st.absmax, 64, "nf4", None, None, None, None)
            for knob in (1101, 1202, 1304, 1401, 2000, 2100, 2202, 4000, 4002, 4003, 5000, 6000):  # (50: streaming MFMA kernel, row passes above 16 rows; 60: the tall-tile experiment)
                bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, knob)
                try:
                    d = repeat(lambda: [hip._gemm_4bit_fused(x, q, *args, kernel=2)], f"forced {knob} {N} x {K} M = {M} nested {int(dq)}")
                    fam = "forced " + FAMILY.get(bnb.lib.bnb_mi355x_last_gemm_kernel(), "?")
                finally:
                    bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
                c = seen.setdefault(fam, [0, 0])
                c[0] += 1
                c[1] += d
    # round 6: matrices with fewer tiles than CUs (routed to the streaming MFMA kernel by a table) and groups that share x as ONE launch
    for (N, K) in ((1376, 4096), (512, 4096), (2560, 2560), (1376, 2752)):
        torch.manual_seed(N + K)
        W = (torch.randn(N, K, device=DEV) / K**0.5).bfloat16()
        q, st = F.quantize_4bit(W, blocksize=64, quant_type="nf4")
        for M in (2, 4, 8, 16):
            x = torch.randn(M, K, device=DEV).bfloat16()
            d = repeat(lambda: [hip._gemm_4bit_fused(x, q, st.shape, st.absmax, 64, "nf4", None, None, None, None)], f"small matrix {N} x {K} M = {M}")
            c = seen.setdefault("small " + FAMILY.get(bnb.lib.bnb_mi355x_last_gemm_kernel(), "?"), [0, 0])
            c[0] += 1
            c[1] += d
    for heights, K, dq in (((4096,) * 4, 4096, False), ((4096, 1024, 1024), 4096, True), ((11008, 11008), 4096, False), ((512,) * 3, 4096, False)):
        torch.manual_seed(len(heights) + K)
        qs, sts = [], []
        for N in heights:
            W = (torch.randn(N, K, device=DEV) / K**0.5).bfloat16()
            q, st = F.quantize_4bit(W, blocksize=64, quant_type="nf4", compress_statistics=dq)
            qs.append(q)
            sts.append(st)
        for M in (1, 2, 4, 16, 32):  # (32 rows: row passes of the grouped launch where the group is small enough)
            x = torch.randn(M, K, device=DEV).bfloat16()
            d = repeat(lambda: bnb.matmul_4bit_grouped(x, qs, sts), f"group {heights} M = {M} nested {int(dq)}")
            c = seen.setdefault("group " + FAMILY.get(bnb.lib.bnb_mi355x_last_gemm_kernel(), "?"), [0, 0])
            c[0] += 1
            c[1] += d
    for fam, (cases, diff) in seen.items():
        print(f"{fam:12s} {cases:3d} cases x {a.runs} runs: {diff} differing runs" + ("   <-- FAIL" if diff else ""), flush=True)

    # standalone kernels
    W = (torch.randn(4096, 4096, device=DEV) * 0.03).bfloat16()
    for qt, bs in (("nf4", 64), ("fp4", 128), ("nf4", 32)):
        def nested():
            p_, s_ = F.quantize_4bit(W, blocksize=bs, quant_type=qt, compress_statistics=True)
            return [p_, s_.absmax, s_.state2.absmax, s_.offset.reshape(1)]

        d = repeat(nested, f"quantize_4bit nested {qt} bs {bs}")
        q, st = F.quantize_4bit(W, blocksize=bs, quant_type=qt, compress_statistics=True)
        d += repeat(lambda: [F.dequantize_4bit(q, st)], f"dequantize_4bit nested {qt} bs {bs}")
        q2, st2 = F.quantize_4bit(W, blocksize=bs, quant_type=qt)
        d += repeat(lambda: [F.dequantize_4bit(q2, st2)], f"dequantize_4bit {qt} bs {bs}")
        print(f"quantize_4bit (one-call nested) / dequantize_4bit {qt} bs {bs}: {d} differing runs" + ("   <-- FAIL" if d else ""), flush=True)
    am = torch.rand(1 << 22, device=DEV) + 0.5
    d = repeat(lambda: list(torch.ops.bitsandbytes.quantize_blockwise.default(am, F._dynamic_map(am.device), 256)), "quantize_blockwise 4 M")
    print(f"quantize_blockwise (byte-table encoder): {d} differing runs" + ("   <-- FAIL" if d else ""), flush=True)
    print("DETERMINISM_STRESS " + ("OK" if bad_total == 0 else f"FAILED ({bad_total})"))
    sys.exit(0 if bad_total == 0 else 1)


if __name__ == "__main__":
    main()
