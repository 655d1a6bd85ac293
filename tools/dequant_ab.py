#!/usr/bin/env python3
"""Round 5 A/B of dequantize4 (csrc/dequantize4.hip): outputs per workgroup (packed dwords per lane 1 / 2 / 4 / 8 - i.e. how many
rounds of workgroups a tensor is; the shipped 4 makes a 4096^2 tensor exactly ONE round of 8 resident workgroups per CU, so all
loads are requested at once and all stores drain at once) and, for fp32 outputs, the line-contiguous lane mapping (one 16-byte store
per unit instead of two half-line stores per lane). C ABI on pre-allocated buffers, graph of R launches over R distinct tensors
(> 512 MiB), regions of >= 10 ms, round-robin over the variants, median of 5. First: every variant's output equals the shipped
kernel's bit for bit. Floors beside it: filling / copying the same output tensors with torch (a write-only and a read + write stream).
    python tools/dequant_ab.py"""
import ctypes as ct
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bitsandbytes_amd as bnb  # noqa: E402

lib = bnb.lib
DT = {torch.float32: "fp32", torch.float16: "fp16", torch.bfloat16: "bf16"}


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
    for n, dt, bs in ((4096 * 4096, torch.bfloat16, 64), (8192 * 4096, torch.bfloat16, 64), (11008 * 4096, torch.bfloat16, 64), (8192 * 8192, torch.bfloat16, 64),
                      (2048 * 2048, torch.bfloat16, 64), (4096 * 4096, torch.float16, 128), (4096 * 4096, torch.float32, 64), (8192 * 8192, torch.float32, 64), (2048 * 2048, torch.float32, 64)):
        es = torch.finfo(dt).bits // 8
        per = n * es + n // 2 + 4 * (n // bs)
        R = max(4, int(600e6 // per) + 1)
        g0 = torch.Generator(device="cuda").manual_seed(0)
        packed = [torch.randint(0, 256, (n // 2,), device="cuda", dtype=torch.uint8, generator=g0) for _ in range(R)]
        absmax = [torch.rand(n // bs, device="cuda", generator=g0) + 0.5 for _ in range(R)]
        outs = [torch.empty(n, device="cuda", dtype=dt) for _ in range(R)]
        other = [torch.empty(n, device="cuda", dtype=dt) for _ in range(2)]
        deq = getattr(lib, f"cdequantize_blockwise_{DT[dt]}_nf4")
        variants = [("u = 4", 14), ("u = 2", 12), ("u = 8", 18), ("u = 16", 26), ("built-in", 0)]
        if dt == torch.float32:
            variants += [("lines u = 2", 32), ("lines u = 4", 34), ("lines u = 8", 38), ("general (round 4)", 40)]

        def run(knob):
            def fn(s):
                lib.bnb_mi355x_set_tuning(knob, 0, 0, 0)
                for i in range(R):
                    deq(None, ptr(packed[i]), ptr(absmax[i]), ptr(outs[i]), bs, n, ct.c_void_p(s.cuda_stream))
                lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
            return fn

        # bit identity
        run(0)(torch.cuda.current_stream())
        torch.cuda.synchronize()
        ref = outs[0].clone()
        bad = []
        for name, knob in variants[1:]:
            outs[0].zero_()
            run(knob)(torch.cuda.current_stream())
            torch.cuda.synchronize()
            if not torch.equal(outs[0].view(torch.int16 if es == 2 else torch.int32), ref.view(torch.int16 if es == 2 else torch.int32)):
                bad.append(name)
        graphs = [capture(run(knob)) for _, knob in variants]
        graphs.append(capture(lambda s: [o.zero_() for o in outs]))
        graphs.append(capture(lambda s: [o.copy_(other[i & 1]) for i, o in enumerate(outs)]))
        names = [v[0] for v in variants] + ["torch fill", "torch copy"]
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
        print(f"# n = {n} ({DT[dt]}, bs {bs}), {R} tensors; algorithmic {per / 1e6:.1f} MB; bit identity: {'all identical' if not bad else 'DIFFERENT: ' + str(bad) + '  <-- FAIL'}")
        for nm, m in zip(names, med):
            bytes_ = per if not nm.startswith("torch") else (n * es if nm == "torch fill" else 2 * n * es)
            print(f"   {nm:14s} {m:8.2f} us   {bytes_ / m / 1e6:6.2f} TB/s   {bytes_ / m / 1e3 / 80:5.1f} % of 8 TB/s", flush=True)
        del packed, absmax, outs, other, graphs
        torch.cuda.empty_cache()


def main8():
    """dequantize_blockwise (8-bit, dynamic map, blocksize 256, fp32 out): four units in flight per lane (shipped) vs one (knob 6)."""
    import bitsandbytes_amd.functional as F

    code = F.create_dynamic_map().cuda()
    for n in (4096 * 4096, 8192 * 8192, 262144):
        per = n + 4 * n + 4 * (n // 256)
        R = max(4, min(512, int(600e6 // per) + 1))
        g0 = torch.Generator(device="cuda").manual_seed(0)
        q8 = [torch.randint(0, 256, (n,), device="cuda", dtype=torch.uint8, generator=g0) for _ in range(R)]
        am = [torch.rand(n // 256, device="cuda", generator=g0) + 0.5 for _ in range(R)]
        outs = [torch.empty(n, device="cuda") for _ in range(R)]

        def run(knob):
            def fn(s):
                lib.bnb_mi355x_set_tuning(knob, 0, 0, 0)
                for i in range(R):
                    lib.cdequantize_blockwise_fp32(ptr(code), ptr(q8[i]), ptr(am[i]), ptr(outs[i]), 256, n, ct.c_void_p(s.cuda_stream))
                lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
            return fn

        run(0)(torch.cuda.current_stream())
        torch.cuda.synchronize()
        ref = outs[0].clone()
        outs[0].zero_()
        run(6)(torch.cuda.current_stream())
        torch.cuda.synchronize()
        same = torch.equal(outs[0].view(torch.int32), ref.view(torch.int32))
        graphs = [capture(run(0)), capture(run(6))]
        t0 = timed(*graphs[0], 3)
        reps = max(3, int(10000.0 / t0) + 1)
        samples = [[], []]
        for r in range(5):
            for i in ((0, 1) if r % 2 == 0 else (1, 0)):
                samples[i].append(timed(*graphs[i], reps) / R)
        med = [statistics.median(x) for x in samples]
        print(f"# dequantize 8-bit -> fp32, n = {n}, {R} tensors; algorithmic {per / 1e6:.2f} MB; bit identity: {'identical' if same else 'DIFFERENT  <-- FAIL'}")
        for nm, m in zip(("4 units (shipped)", "1 unit (round 1-4)"), med):
            print(f"   {nm:20s} {m:8.2f} us   {per / m / 1e6:6.2f} TB/s   {per / m / 1e3 / 80:5.1f} % of 8 TB/s", flush=True)
        del q8, am, outs, graphs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
    main8()
