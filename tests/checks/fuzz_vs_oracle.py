#!/usr/bin/env python3
"""Random-shape fuzz of the public operators against the oracle (TEST INFRASTRUCTURE: imports oracle/): the parametrised tests pin
chosen shapes; this draws them - ragged N, K at every multiple of the blocksize, every M band of the router, every blocksize / type /
statistics kind / bias - and checks, per draw:
  * quantize_4bit: packed bytes and absmax bit-exact (nested: given the device's offset, codes and second-level absmax bit-exact);
  * dequantize_4bit: equal to the oracle's values (modulo the two documented quirks of conftest.same_values_ftz);
  * gemm_4bit (routed kernel, the public op): relative Frobenius error <= 1e-2 (16-bit) / 1e-5 (fp32) against fp32 dequantize + fp32 linear.
    python tests/checks/fuzz_vs_oracle.py [--draws 300] [--seed 1]"""
import argparse
import os
import random
import sys
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from conftest import rel_err, same_values_ftz  # noqa: E402
from oracle import oracle as O  # noqa: E402

DEV = "cuda"
FAMILY = {1: "stream", 2: "generic", 3: "rt", 4: "pc", 6: "kq", 7: "sm", 0: "unfused"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--draws", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = random.Random(a.seed)
    torch.manual_seed(a.seed)
    fails = 0
    fam_count = {}
    worst = {}
    for d in range(a.draws):
        bs = rng.choice([32, 64, 64, 64, 128, 128, 256, 512, 1024])
        qt = rng.choice(["nf4", "nf4", "fp4"])
        dt = rng.choice([torch.bfloat16, torch.bfloat16, torch.float16, torch.float32])
        dq = rng.random() < 0.5
        # K: a multiple of the blocksize; half the draws a multiple of 256 (the MFMA kernels' precondition), long and short rows
        if rng.random() < 0.5:
            K = 256 * rng.randint(1, 44)
            K = max(K, bs) // bs * bs
        else:
            K = bs * rng.randint(1, max(1, 4096 // bs))
        M = rng.choice([1, 1, 2, 3, 4, 5, 7, 8, 12, 16, 17, 24, 32, 33, 40, 48, 49, 64, 65, 96, 130, 200, 300, 513, 600])
        n_cap = max(1, int(1.5e8 // (M * K)))
        N = min(rng.choice([1, 7, 16, 96, 130, 200, 256, 384, 1000, 1376, 2048, 4096, rng.randint(1, 3000)]), n_cap)
        if rng.random() < 0.04 and K % 256 == 0:  # a few draws large enough for the K-quarter kernel's range (>= 16 M weights, M >= 17)
            M, N, K = rng.choice([17, 33, 64, 100]), 4096 + 16 * rng.randint(0, 8), max(4096 // bs * bs, bs)
        if rng.random() < 0.08 and bs >= 64 and dt != torch.float32:
            # a few draws in the streaming MFMA kernel's range (round 6): 2 ... 16 rows (to 64 where K is not a multiple of 256) on a
            # matrix of >= 3072 rows, any K % 64 == 0 that the blocksize divides
            N = 3072 + rng.randint(0, 1200)
            K = max(bs, 64) * rng.randint(1, max(1, 3072 // max(bs, 64)))
            M = rng.choice([2, 3, 4, 5, 8, 9, 13, 16]) if K % 256 == 0 else rng.choice([2, 7, 16, 17, 40, 64])
        bias = rng.random() < 0.4
        label = f"draw {d}: M {M} N {N} K {K} bs {bs} {qt} {str(dt)[6:]} nested {int(dq)} bias {int(bias)}"
        try:
            W = (torch.randn(N, K) / K**0.5).to(dt)
            if rng.random() < 0.2:
                W[rng.randrange(N)] = 0  # an all-zero row: all-zero blocks
            x = torch.randn(M, K).to(dt)
            b = torch.randn(N).to(dt) if bias else None
            q, st = F.quantize_4bit(W.to(DEV), blocksize=bs, quant_type=qt, compress_statistics=dq)
            q_o, am_o = O.quantize_4bit(W, bs, qt)
            assert torch.equal(q.cpu(), q_o), "packed bytes"
            if dq:
                off = st.offset.cpu()
                code = st.state2.code.cpu()
                q8_o, am2_o = O.quantize_blockwise(am_o - off, code, 256)
                assert torch.equal(st.absmax.cpu(), q8_o) and torch.equal(st.state2.absmax.cpu(), am2_o), "nested statistics"
                am_rec = O.dequantize_blockwise(q8_o, am2_o, code, 256, torch.float32) + off
                kw = dict(absmax_8bit=st.absmax, absmax_code=st.state2.code, absmax_offset=st.offset)
                am_dev = st.state2.absmax
            else:
                assert torch.equal(st.absmax.cpu(), am_o), "absmax"
                am_rec, kw, am_dev = am_o, {}, st.absmax
            assert same_values_ftz(F.dequantize_4bit(q, st).cpu(), O.dequantize_4bit(q_o, am_rec, bs, qt, W.shape, dt)), "dequantize"
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                y = torch.ops.bitsandbytes.gemm_4bit.default(x.to(DEV), q, st.shape, am_dev, bs, qt, bias=None if b is None else b.to(DEV), **kw)
            from bitsandbytes_amd.backends import hip

            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                routed = hip._gemm_4bit_route(dt, M, N, K, bs, dq)
            fam = FAMILY.get(bnb.lib.bnb_mi355x_last_gemm_kernel() if routed == "fused" else 0, "?")
            y_o = O.gemm_4bit(x, q_o, (N, K), am_rec, bs, qt, b)[1]
            e = rel_err(y.float().cpu(), y_o)
            tol = 1e-5 if dt == torch.float32 else 1e-2
            fam_count[fam] = fam_count.get(fam, 0) + 1
            worst[fam] = max(worst.get(fam, 0.0), e if dt != torch.float32 else 0.0)
            assert e < tol, f"gemm rel err {e:.3e} (family {fam})"
        except Exception as exc:  # noqa: BLE001
            fails += 1
            print(f"{label}: FAIL {type(exc).__name__}: {exc}", flush=True)
    print(f"{a.draws} draws, seed {a.seed}: {fails} failures; kernel families hit: " +
          ", ".join(f"{k} {v} (worst 16-bit err {worst[k]:.1e})" for k, v in sorted(fam_count.items())))
    print("FUZZ " + ("OK" if fails == 0 else f"FAILED ({fails})"))
    sys.exit(0 if fails == 0 else 1)


if __name__ == "__main__":
    main()
