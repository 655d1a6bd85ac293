#!/usr/bin/env python3
"""Run-to-run reproducibility stress of the pre-scaled-operand MFMA kernel (tuning cfg 30): every shape / format 40 times,
outputs compared bit for bit with the first run and against dequantize_4bit + an fp64 product."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitsandbytes_amd as bnb  # noqa: E402
import bitsandbytes_amd.functional as F  # noqa: E402
from bitsandbytes_amd.backends import hip  # noqa: E402

torch.manual_seed(0)
print(os.environ.get("BNB_MI355X_LIBRARY", "default library"))
RUNS = int(os.environ.get("STRESS_RUNS", "40"))
NESTED = os.environ.get("STRESS_NESTED", "1") == "1"   # (nested absmax is not routed to this kernel: run through hip._gemm_4bit_fused it takes the fallback)
for (M, N, K) in ((5, 256, 1024), (40, 260, 1024), (64, 512, 4096), (128, 1376, 4096), (16, 200, 2048), (17, 512, 10752), (200, 256, 256),
                  (300, 384, 1024), (64, 8192, 8192)):
    for dtype, qt, bs, dq in ((torch.bfloat16, "nf4", 64, False), (torch.bfloat16, "nf4", 64, True),
                              (torch.float16, "fp4", 128, True), (torch.float16, "nf4", 256, False)):
        if dq and not NESTED:
            continue
        W = (torch.randn(N, K) / K**0.5).to(dtype).cuda()
        x = torch.randn(M, K).to(dtype).cuda()
        bias = torch.randn(N).to(dtype).cuda()
        q, st = F.quantize_4bit(W, blocksize=bs, quant_type=qt, compress_statistics=dq)
        ref = x.double() @ F.dequantize_4bit(q, st).double().T + bias.double()
        bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 3000)
        try:
            if dq:
                ys = [hip._gemm_4bit_fused(x, q, st.shape, st.state2.absmax, st.blocksize, st.quant_type, bias, st.absmax, st.state2.code,
                                           st.offset, kernel=2) for _ in range(RUNS)]
            else:
                ys = [hip._gemm_4bit_fused(x, q, st.shape, st.absmax, st.blocksize, st.quant_type, bias, None, None, None, kernel=2)
                      for _ in range(RUNS)]
        finally:
            bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
        torch.cuda.synchronize()
        nd = sum(int(not torch.equal(ys[0], y)) for y in ys[1:])
        err = max(((y.double() - ref).abs().max() / ref.abs().max()).item() for y in ys)
        print(f"M={M:4d} N={N:5d} K={K:5d} {str(dtype)[6:]:9s} {qt} bs={bs:3d} nested={int(dq)}: runs differing from the first {nd:3d}/{RUNS - 1}, "
              f"worst rel err {err:.2e}", flush=True)
