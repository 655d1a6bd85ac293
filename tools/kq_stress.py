#!/usr/bin/env python3
"""Repeated launches of the K-quarter MFMA kernel (tuning cfg 40) on small and large shapes: every result must equal the first one
bit for bit and stay within tolerance of a fp32 dequantize + matmul; a memory fault or a differing run is a bug.
    python tools/kq_stress.py [runs]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from rt_variant_ab import one  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
print(torch.cuda.get_device_name(0), os.environ.get("BNB_MI355X_LIBRARY", "default library"), flush=True)
for (N, K, M, dq, qt, bs, dt) in ((256, 512, 64, False, "nf4", 64, torch.bfloat16), (256, 256, 17, True, "nf4", 64, torch.bfloat16),
                                  (384, 1024, 33, False, "nf4", 64, torch.bfloat16), (1000, 2816, 64, True, "nf4", 64, torch.bfloat16),
                                  (130, 512, 17, False, "fp4", 128, torch.float16), (2048, 4096, 48, True, "nf4", 64, torch.bfloat16),
                                  (8192, 8192, 64, True, "nf4", 64, torch.bfloat16), (4096, 11008 - 11008 % 256, 64, False, "nf4", 64, torch.bfloat16)):
    g = torch.Generator(device="cuda").manual_seed(N + K + M)
    W = (torch.randn(N, K, device="cuda", generator=g) / K**0.5).to(dt)
    q, st = F.quantize_4bit(W, blocksize=bs, quant_type=qt, compress_statistics=dq)
    x = torch.randn(M, K, device="cuda", generator=g).to(dt)
    ref = x.float() @ F.dequantize_4bit(q, st).float().t()
    for knob in (4000, 4002):
        bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, knob)
        try:
            y0 = one(q, st, x).clone()
            err = float((y0.float() - ref).norm() / ref.norm())
            differing = 0
            for _ in range(runs):
                differing += int(not torch.equal(one(q, st, x), y0))
            torch.cuda.synchronize()
        finally:
            bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
        print(f"{N:5d} x {K:5d} M = {M:3d} {qt} bs {bs} nested {int(dq)} knob {knob}: err {err:.2e}, {differing} of {runs} runs differ"
              + ("" if err < 1e-2 and differing == 0 else "   <-- FAIL"), flush=True)
