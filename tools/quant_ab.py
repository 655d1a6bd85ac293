#!/usr/bin/env python3
"""Round 5 A/B of quantize4 (csrc/quantize4.hip): 2048-element chunks per workgroup (2 = shipped on large NF4 inputs since this A/B / 4 = round 4 / 8) - the same
question tools/dequant_ab.py asked of dequantize4: how many workgroups a tensor should be. C ABI on pre-allocated buffers, graph of R
launches over R distinct tensors (> 512 MiB), regions of >= 10 ms, round-robin, median of 5; first every variant's packed codes and
absmax equal the shipped kernel's bit for bit. Beside it: a torch copy of the input to a tensor of the output's size class is not a
fair floor (the op reads 4 x what it writes), so the floor is a plain read of the input: `x.sum()` is NOT used (it is its own kernel
with its own launches) - the line to compare with is dequantize4's u = 8 (the same bytes the other way round).
    python tools/quant_ab.py"""
import ctypes as ct
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bitsandbytes_amd as bnb  # noqa: E402

lib = bnb.lib
DT = {torch.float32: (0, "fp32"), torch.float16: (1, "fp16"), torch.bfloat16: (2, "bf16")}
QT = {"fp4": 1, "nf4": 2}


def ptr(t):
    return ct.c_void_p(t.data_ptr())


def capture(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(s)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn(s)
        g.replay()
        s.synchronize()
    return g, s


def timed(g, s, reps):
    with torch.cuda.stream(s):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            g.replay()
        e1.record(s)
        s.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    print(torch.cuda.get_device_name(0), lib.bnb_mi355x_version().decode())
    for n, dt, qt, bs in ((4096 * 4096, torch.bfloat16, "nf4", 64), (4096 * 4096, torch.bfloat16, "fp4", 64), (8192 * 8192, torch.bfloat16, "nf4", 64),
                          (4096 * 4096, torch.float16, "nf4", 128), (4096 * 4096, torch.float32, "nf4", 64), (2048 * 2048, torch.bfloat16, "nf4", 64)):
        es = torch.finfo(dt).bits // 8
        per = n * es + n // 2 + 4 * (n // bs)
        R = max(4, int(600e6 // per) + 1)
        src = [torch.randn(n, device="cuda", dtype=torch.float32).to(dt) for _ in range(R)]
        packed = [torch.empty(n // 2, device="cuda", dtype=torch.uint8) for _ in range(R)]
        absmax = [torch.empty(n // bs, device="cuda", dtype=torch.float32) for _ in range(R)]
        variants = [("built-in", 0), ("2 chunks (forced)", 8), ("4 chunks (round 4)", 4), ("8 chunks", 5), ("one tile (FP4 A/B)", 3)]

        def run(knob):
            def fn(s):
                lib.bnb_mi355x_set_tuning(knob, 0, 0, 0)
                for i in range(R):
                    lib.bnb_mi355x_quantize_4bit(ptr(src[i]), DT[dt][0], ptr(absmax[i]), ptr(packed[i]), bs, n, QT[qt], ct.c_void_p(s.cuda_stream))
                lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
            return fn

        run(0)(torch.cuda.current_stream())
        torch.cuda.synchronize()
        ref_p, ref_a = packed[0].clone(), absmax[0].clone()
        bad = []
        for name, knob in variants[1:]:
            packed[0].zero_()
            absmax[0].zero_()
            run(knob)(torch.cuda.current_stream())
            torch.cuda.synchronize()
            if not (torch.equal(packed[0], ref_p) and torch.equal(absmax[0].view(torch.int32), ref_a.view(torch.int32))):
                bad.append(name)
        graphs = [capture(run(knob)) for _, knob in variants]
        t0 = timed(*graphs[0], 3)
        reps = max(3, int(10000.0 / t0) + 1)
        samples = [[] for _ in graphs]
        for r in range(5):
            order = list(range(len(graphs)))
            if r % 2:
                order.reverse()
            for i in order:
                samples[i].append(timed(*graphs[i], reps) / R)
        med = [statistics.median(x) for x in samples]
        print(f"# n = {n} ({DT[dt][1]}, {qt}, bs {bs}), {R} tensors; algorithmic {per / 1e6:.1f} MB; bit identity: {'all identical' if not bad else 'DIFFERENT: ' + str(bad) + '  <-- FAIL'}")
        for (nm, _), m in zip(variants, med):
            print(f"   {nm:20s} {m:8.2f} us   {per / m / 1e6:6.2f} TB/s   {per / m / 1e3 / 80:5.1f} % of 8 TB/s", flush=True)
        del src, packed, absmax, graphs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
