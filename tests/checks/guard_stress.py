#!/usr/bin/env python3
"""Out-of-bounds WRITES and alignment assumptions of every C-ABI entry point of the path: all operands are carved out of larger
buffers at RANDOM element offsets (so pointers are aligned to the element size and nothing more), every output (and scratch) region
sits between guard bands filled with 0xA5, and after each call
  * the guard bands must be untouched,
  * the result must equal (quantize / dequantize: bit for bit; matmuls: within 1e-2 / 1e-5) what the public operator gives on
    freshly allocated, aligned tensors.
Random ragged shapes. (Out-of-bounds READS cannot be seen this way; the kernels' buffer descriptors clamp those.)
    python tests/checks/guard_stress.py [--draws 120] [--seed 1]"""
import argparse
import os
import random
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from bitsandbytes_amd.backends import hip  # noqa: E402

DEV = "cuda"
G = 8192  # guard bytes on each side
PAT = 0xA5
DT_CODE = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
DT_NAME = {torch.float32: "fp32", torch.float16: "fp16", torch.bfloat16: "bf16"}
QT_CODE = {"fp4": 1, "nf4": 2}
lib = bnb.lib


class Carved:
    """nbytes inside a guarded buffer, starting `off` bytes past a 256-byte-aligned address."""

    def __init__(self, nbytes, off):
        self.buf = torch.full((nbytes + 2 * G + 512,), PAT, dtype=torch.uint8, device=DEV)
        base = self.buf.data_ptr()
        self.start = G + (-base - G) % 256 + off
        self.nbytes = nbytes
        self.view = self.buf[self.start:self.start + nbytes]

    def ptr(self):
        return self.view.data_ptr()

    def put(self, t):
        self.view.copy_(t.contiguous().view(-1).view(torch.uint8))
        return self

    def as_tensor(self, dtype, shape):
        return self.view.view(dtype).view(*shape)  # (only when the offset keeps torch's alignment rule for the dtype)

    def bytes(self):
        return self.view.clone()

    def guards_ok(self):
        lo = self.buf[:self.start]
        hi = self.buf[self.start + self.nbytes:]
        return bool((lo == PAT).all()) and bool((hi == PAT).all())


def carve_in(t, rng, elt=None):
    elt = elt or t.element_size()
    c = Carved(t.numel() * t.element_size(), elt * rng.randint(0, 31))
    return c.put(t)


def carve_out(nbytes, rng, elt):
    return Carved(nbytes, elt * rng.randint(0, 31))


def read(c, dtype, n):
    return c.bytes().view(dtype)[:n] if c.nbytes else torch.empty(0, dtype=dtype, device=DEV)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--draws", type=int, default=120)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = random.Random(a.seed)
    torch.manual_seed(a.seed)
    stream = lambda: torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())  # noqa: E731
    fails = {}
    counts = {}

    def check(name, cond, label):
        counts[name] = counts.get(name, 0) + 1
        if not cond:
            fails[name] = fails.get(name, 0) + 1
            print(f"{name}: FAIL {label}", flush=True)

    code8 = F._dynamic_map(torch.device(DEV, torch.cuda.current_device()))
    for d in range(a.draws):
        bs = rng.choice([32, 64, 64, 128, 256, 1024])
        qt = rng.choice(["nf4", "fp4"])
        dt = rng.choice([torch.bfloat16, torch.float16, torch.float32])
        es = torch.empty(0, dtype=dt).element_size()
        # ---------------------------------------------------------------- quantize_4bit / nested / dequantize (flat, ragged n)
        n = rng.choice([1, 5, 63, 64, 65, 1000, 4096, 64 * 255 + 7, 8192 * 3 + 1, rng.randint(1, 300000)])
        X = (torch.randn(n, device=DEV) * 0.1).to(dt)
        blocks = -(n // -bs)
        label = f"draw {d}: n {n} bs {bs} {qt} {DT_NAME[dt]}"
        q_ref, am_ref = torch.ops.bitsandbytes.quantize_4bit.default(X, bs, qt, torch.uint8)
        cx = carve_in(X, rng)
        c_am, c_q = carve_out(blocks * 4, rng, 4), carve_out((n + 1) // 2, rng, 1)
        lib.bnb_mi355x_quantize_4bit(cx.ptr(), DT_CODE[dt], c_am.ptr(), c_q.ptr(), bs, n, QT_CODE[qt], stream())
        torch.cuda.synchronize()
        check("quantize_4bit", c_am.guards_ok() and c_q.guards_ok() and torch.equal(read(c_q, torch.uint8, (n + 1) // 2), q_ref.view(-1))
              and torch.equal(read(c_am, torch.int32, blocks), am_ref.view(torch.int32)), label)
        # nested: one call
        _, st = F.quantize_4bit(X, blocksize=bs, quant_type=qt, compress_statistics=True)
        c_q2, c_scr = carve_out((n + 1) // 2, rng, 1), carve_out((blocks + 1536) * 4, rng, 4)
        c_a8, c_a2, c_off = carve_out(blocks, rng, 1), carve_out(-(blocks // -256) * 4, rng, 4), carve_out(4, rng, 4)
        c_code = carve_in(code8, rng)
        lib.bnb_mi355x_quantize_4bit_nested(cx.ptr(), DT_CODE[dt], n, bs, QT_CODE[qt], c_q2.ptr(), c_scr.ptr(), c_code.ptr(), c_a8.ptr(),
                                            c_a2.ptr(), c_off.ptr(), stream())
        torch.cuda.synchronize()
        ok = all(c.guards_ok() for c in (c_q2, c_scr, c_a8, c_a2, c_off))
        ok = ok and torch.equal(read(c_q2, torch.uint8, (n + 1) // 2), q_ref.view(-1)) and torch.equal(read(c_a8, torch.uint8, blocks), st.absmax)
        ok = ok and torch.equal(read(c_a2, torch.int32, -(blocks // -256)), st.state2.absmax.view(torch.int32))
        ok = ok and torch.equal(read(c_off, torch.int32, 1), st.offset.view(torch.int32).reshape(1))
        check("quantize_4bit_nested", ok, label)
        # dequantize, plain and nested statistics
        want = torch.ops.bitsandbytes.dequantize_4bit.default(q_ref, am_ref, bs, qt, [n], dt)
        c_o = carve_out(n * es, rng, es)
        c_qi, c_ami = carve_in(q_ref.view(-1), rng), carve_in(am_ref, rng)
        getattr(lib, f"cdequantize_blockwise_{DT_NAME[dt]}_{qt}")(None, c_qi.ptr(), c_ami.ptr(), c_o.ptr(), bs, n, stream())
        torch.cuda.synchronize()
        check("dequantize_4bit", c_o.guards_ok() and torch.equal(read(c_o, torch.uint8, n * es), want.view(-1).view(torch.uint8)), label)
        want_n = F.dequantize_4bit(q_ref, st)
        c_o2 = carve_out(n * es, rng, es)
        c_a8i, c_a2i, c_offi = carve_in(st.absmax, rng), carve_in(st.state2.absmax, rng), carve_in(st.offset.reshape(1), rng)
        lib.bnb_mi355x_dequantize_4bit_nested(DT_CODE[dt], c_qi.ptr(), c_a8i.ptr(), c_a2i.ptr(), c_code.ptr(), c_offi.ptr(), c_o2.ptr(), bs, n,
                                              QT_CODE[qt], stream())
        torch.cuda.synchronize()
        check("dequantize_4bit_nested", c_o2.guards_ok() and torch.equal(read(c_o2, torch.uint8, n * es), want_n.reshape(-1).view(torch.uint8)), label)
        # ---------------------------------------------------------------- 8-bit blockwise pair
        bs8 = rng.choice([64, 256, 256, 4096])
        n8 = rng.choice([1, 255, 256, 1000, 70000, rng.randint(1, 1 << 21)])
        X8 = torch.randn(n8, device=DEV).to(dt)
        q8_ref, am8_ref = torch.ops.bitsandbytes.quantize_blockwise.default(X8, code8, bs8)
        cx8 = carve_in(X8, rng)
        c_am8, c_q8 = carve_out(-(n8 // -bs8) * 4, rng, 4), carve_out(n8, rng, 1)
        lib.bnb_mi355x_quantize_8bit(c_code.ptr(), cx8.ptr(), DT_CODE[dt], c_am8.ptr(), c_q8.ptr(), bs8, n8, stream())
        torch.cuda.synchronize()
        check("quantize_blockwise", c_am8.guards_ok() and c_q8.guards_ok() and torch.equal(read(c_q8, torch.uint8, n8), q8_ref.view(-1))
              and torch.equal(read(c_am8, torch.int32, -(n8 // -bs8)), am8_ref.view(torch.int32)), f"draw {d}: n {n8} bs {bs8} {DT_NAME[dt]}")
        want8 = torch.ops.bitsandbytes.dequantize_blockwise.default(q8_ref, am8_ref, code8, bs8, dt)
        c_o8 = carve_out(n8 * es, rng, es)
        c_q8i, c_am8i = carve_in(q8_ref.view(-1), rng), carve_in(am8_ref, rng)
        getattr(lib, f"cdequantize_blockwise_{DT_NAME[dt]}")(c_code.ptr(), c_q8i.ptr(), c_am8i.ptr(), c_o8.ptr(), bs8, n8, stream())
        torch.cuda.synchronize()
        check("dequantize_blockwise", c_o8.guards_ok() and torch.equal(read(c_o8, torch.uint8, n8 * es), want8.view(-1).view(torch.uint8)),
              f"draw {d}: n {n8} bs {bs8} {DT_NAME[dt]}")
        # ---------------------------------------------------------------- gemm_4bit, gemm_4bit_grad_input
        K = bs * rng.randint(1, max(1, 2048 // bs)) if rng.random() < 0.5 else max(256 * rng.randint(1, 16) // bs * bs, bs)
        M = rng.choice([1, 1, 2, 3, 4, 5, 8, 16, 17, 33, 48, 64, 65, 130, 300])
        N = rng.choice([1, 7, 16, 96, 130, 256, 1000, 1376, rng.randint(1, 1500)])
        nested = rng.random() < 0.5
        W = (torch.randn(N, K, device=DEV) / K**0.5).to(dt)
        q, stg = F.quantize_4bit(W, blocksize=bs, quant_type=qt, compress_statistics=nested)
        x = torch.randn(M, K, device=DEV).to(dt)
        bias = torch.randn(N, device=DEV).to(dt) if rng.random() < 0.5 else None
        label = f"draw {d}: M {M} N {N} K {K} bs {bs} {qt} {DT_NAME[dt]} nested {int(nested)} bias {int(bias is not None)}"
        if nested:
            args = (stg.state2.absmax, stg.absmax, stg.state2.code, stg.offset.reshape(1))
        else:
            args = (stg.absmax, None, None, None)
        fused = dt != torch.float32 and M <= 512 or M <= 4
        if fused:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                want = hip._gemm_4bit_fused(x, q, stg.shape, args[0], bs, qt, bias, args[1], args[2], None if args[3] is None else stg.offset)
            cA, cB, cAm = carve_in(x, rng), carve_in(q.view(-1), rng), carve_in(args[0], rng)
            cA8 = carve_in(args[1], rng) if nested else None
            cC2 = carve_in(args[2], rng) if nested else None
            cOf = carve_in(args[3], rng) if nested else None
            cBias = carve_in(bias, rng) if bias is not None else None
            cOut = carve_out(M * N * es, rng, es)
            ws_bytes = lib.bnb_mi355x_gemm_4bit_workspace_bytes(0, DT_CODE[dt], M, N, K, bs)
            cWs = carve_out(ws_bytes, rng, 4) if ws_bytes else None
            p = lambda c: None if c is None else c.ptr()  # noqa: E731
            lib.bnb_mi355x_gemm_4bit(0, DT_CODE[dt], cA.ptr(), cB.ptr(), cAm.ptr(), p(cA8), p(cC2), p(cOf), None, cOut.ptr(), p(cBias), M, N, K,
                                     bs, QT_CODE[qt], p(cWs), ws_bytes, stream())
            torch.cuda.synchronize()
            got = read(cOut, dt, M * N).view(M, N)
            err = float((got.double() - want.double()).norm() / want.double().norm().clamp_min(1e-30))
            check("gemm_4bit", cOut.guards_ok() and (cWs is None or cWs.guards_ok()) and err < (1e-5 if dt == torch.float32 else 1e-2) and
                  bool(torch.isfinite(got).all()), label + f" err {err:.2e}")
        if dt != torch.float32 and M <= 128 and bs >= 64 and lib.bnb_mi355x_gemm_4bit_grad_input_supported(DT_CODE[dt], M, N, K, bs):
            g = torch.randn(M, N, device=DEV).to(dt)
            op = torch.ops.bitsandbytes_amd.gemm_4bit_grad_input.default
            if nested:
                want = op(g, q, stg.shape, stg.state2.absmax, bs, qt, absmax_8bit=stg.absmax, absmax_code=stg.state2.code, absmax_offset=stg.offset)
            else:
                want = op(g, q, stg.shape, stg.absmax, bs, qt)
            # (the fused backward wants 16-byte-aligned grad_out / B - the op falls back otherwise; through the C ABI: aligned carves)
            cG, cB2 = Carved(g.numel() * es, 0).put(g), Carved(q.numel(), 0).put(q.view(-1))
            cAm = carve_in(args[0], rng)
            cA8 = carve_in(args[1], rng) if nested else None
            cC2 = carve_in(args[2], rng) if nested else None
            cOf = carve_in(args[3], rng) if nested else None
            cOut = Carved(M * K * es, 0)
            ws_bytes = lib.bnb_mi355x_gemm_4bit_grad_input_workspace_bytes(M, N, K)
            cWs = Carved(ws_bytes, 0) if ws_bytes else None
            p = lambda c: None if c is None else c.ptr()  # noqa: E731
            lib.bnb_mi355x_gemm_4bit_grad_input(DT_CODE[dt], cG.ptr(), cB2.ptr(), cAm.ptr(), p(cA8), p(cC2), p(cOf), cOut.ptr(), M, N, K, bs,
                                                QT_CODE[qt], p(cWs), ws_bytes, stream())
            torch.cuda.synchronize()
            got = read(cOut, dt, M * K).view(M, K)
            check("gemm_4bit_grad_input", cOut.guards_ok() and (cWs is None or cWs.guards_ok()) and torch.equal(got, want), label)
    print(f"{a.draws} draws, seed {a.seed}: " + ", ".join(f"{k} {v - fails.get(k, 0)}/{v}" for k, v in counts.items()))
    total = sum(fails.values())
    print("GUARD_STRESS " + ("OK" if total == 0 else f"FAILED ({total})"))
    sys.exit(0 if total == 0 else 1)


if __name__ == "__main__":
    main()
