"""Kernel-level timing of the standalone quantize_4bit / dequantize_4bit streams (SURVEY §8 rows a/d).

Calls the C ABI directly on pre-allocated buffers inside a captured hipGraph, rotating over enough
distinct tensors to exceed the 256 MiB Infinity Cache, so the number is the HBM-resident kernel time
without Python / allocator overhead. Algorithmic bytes: quantize reads n*sizeof(T), writes n/2 + 4n/bs;
dequantize the reverse.

    python tools/stream_bench.py [--n 16777216] [--quick]
"""
import argparse
import ctypes as ct
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bitsandbytes_amd as bnb  # noqa: E402

DT = {torch.float32: (0, "fp32"), torch.float16: (1, "fp16"), torch.bfloat16: (2, "bf16")}
QT = {"fp4": 1, "nf4": 2}


def ptr(t):
    return ct.c_void_p(t.data_ptr())


def time_graph(launch, rounds=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        launch(s)  # warm
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            launch(s)
        g.replay()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(rounds):
            g.replay()
        e1.record(s)
        s.synchronize()
    return e0.elapsed_time(e1) / rounds * 1e3  # us per graph


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096 * 4096)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--q4-one-tile", action="store_true", help="force the one-tile form of quantize4_kernel (A/B of the pipelined form)")
    a = ap.parse_args()
    if a.q4_one_tile:
        bnb.lib.bnb_mi355x_set_tuning(3, 0, 0, 0)
    n = a.n
    lib = bnb.lib
    print(f"# n={n} elements; graph of R launches over R distinct tensors (R*bytes > 512 MiB)")
    print(f"{'op':10s} {'dtype':5s} {'qt':3s} {'bs':>5s} {'us':>8s} {'GB/s':>8s} {'frac of 8 TB/s':>8s}")
    combos = [(torch.bfloat16, "nf4", 64), (torch.float16, "nf4", 64), (torch.float32, "nf4", 64),
              (torch.bfloat16, "fp4", 64), (torch.bfloat16, "nf4", 128), (torch.bfloat16, "nf4", 32),
              (torch.bfloat16, "nf4", 256), (torch.bfloat16, "nf4", 4096), (torch.float16, "fp4", 128)]
    if a.quick:
        combos = combos[:4]
    for dt, qt, bs in combos:
        es = torch.finfo(dt).bits // 8
        per = n * es + n // 2 + 4 * (n // bs)
        R = max(4, int(600e6 // per) + 1)
        src = [torch.randn(n, device="cuda", dtype=torch.float32).to(dt) for _ in range(R)]
        packed = [torch.empty(n // 2, device="cuda", dtype=torch.uint8) for _ in range(R)]
        absmax = [torch.empty(n // bs, device="cuda", dtype=torch.float32) for _ in range(R)]
        outs = [torch.empty(n, device="cuda", dtype=dt) for _ in range(R)]
        code = DT[dt][0]
        deq = getattr(lib, f"cdequantize_blockwise_{DT[dt][1]}_{qt}")

        def launch_q(s):
            for i in range(R):
                lib.bnb_mi355x_quantize_4bit(ptr(src[i]), code, ptr(absmax[i]), ptr(packed[i]), bs, n, QT[qt],
                                             ct.c_void_p(s.cuda_stream))

        def launch_d(s):
            for i in range(R):
                deq(None, ptr(packed[i]), ptr(absmax[i]), ptr(outs[i]), bs, n, ct.c_void_p(s.cuda_stream))

        tq = time_graph(launch_q) / R
        td = time_graph(launch_d) / R
        for name, t in (("quantize4", tq), ("dequant4", td)):
            print(f"{name:10s} {DT[dt][1]:5s} {qt:3s} {bs:5d} {t:8.2f} {per / t / 1e3:8.1f} {per / t / 1e3 / 8000:8.3f}")
        del src, packed, absmax, outs
        torch.cuda.empty_cache()
    # 8-bit blockwise pair (the double-quant helper): n elements of fp32 <-> uint8 + absmax per 256
    import bitsandbytes_amd.functional as F
    code = F.create_dynamic_map().cuda()
    n8 = n
    per8 = n8 * 4 + n8 + 4 * (n8 // 256)
    R = max(4, int(600e6 // per8) + 1)
    src = [torch.randn(n8, device="cuda") for _ in range(R)]
    q8 = [torch.empty(n8, device="cuda", dtype=torch.uint8) for _ in range(R)]
    am8 = [torch.empty(n8 // 256, device="cuda") for _ in range(R)]
    out8 = [torch.empty(n8, device="cuda") for _ in range(R)]

    def launch_q8(s):
        for i in range(R):
            lib.bnb_mi355x_quantize_8bit(ptr(code), ptr(src[i]), 0, ptr(am8[i]), ptr(q8[i]), 256, n8, ct.c_void_p(s.cuda_stream))

    def launch_d8(s):
        for i in range(R):
            lib.cdequantize_blockwise_fp32(ptr(code), ptr(q8[i]), ptr(am8[i]), ptr(out8[i]), 256, n8, ct.c_void_p(s.cuda_stream))

    for name, fn in (("quantize8", launch_q8), ("dequant8", launch_d8)):
        t = time_graph(fn) / R
        print(f"{name:10s} {'fp32':5s} {'dyn':3s} {256:5d} {t:8.2f} {per8 / t / 1e3:8.1f} {per8 / t / 1e3 / 8000:8.3f}")


if __name__ == "__main__":
    main()
