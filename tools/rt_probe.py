#!/usr/bin/env python3
"""Debug probe for the register-transposed MFMA kernel: unit-impulse activations read the kernel's effective weight matrix
back column by column and compare it with the dequantized weights (which k / which scale is wrong, if any)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitsandbytes_amd as bnb
import bitsandbytes_amd.functional as F
from bitsandbytes_amd.backends import hip

def main():
    torch.manual_seed(0)
    N, K = 16, 512
    W = (torch.randn(N, K, device="cuda") / K**0.5).bfloat16()
    q, st = F.quantize_4bit(W, blocksize=64, quant_type="nf4")
    Wd = F.dequantize_4bit(q, st).float()                                     # [N, K]
    code = st.code.float()
    qb = q.view(torch.uint8).flatten()
    idx = torch.stack([qb >> 4, qb & 15], 1).flatten().reshape(N, K).long()
    codes = code[idx].bfloat16().float()                                       # T-rounded code values
    scales = st.absmax.reshape(N, K // 64)
    bad = []
    for cfg in (2000, 1100):
        eff = torch.zeros(N, K, device="cuda")
        for k0 in range(0, K, 16):
            x = torch.zeros(16, K, device="cuda", dtype=torch.bfloat16)
            x[torch.arange(16), k0 + torch.arange(16)] = 1.0
            bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, cfg)
            y = hip._gemm_4bit_fused(x, q, st.shape, st.absmax, 64, "nf4", None, None, None, None, kernel=2)
            bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
            eff[:, k0:k0 + 16] = y.float().t()
        ratio = eff / (codes * scales.repeat_interleave(64, 1))
        ok = ((eff - Wd).abs() <= 0.01 * Wd.abs() + 1e-6)
        print(f"cfg {cfg}: {int((~ok).sum())} of {N * K} effective weights differ")
        if (~ok).any():
            n0 = int((~ok).any(1).nonzero()[0])
            ks = (~ok)[n0].nonzero().flatten().tolist()
            print(f"  row {n0}: wrong k = {ks[:64]}{'...' if len(ks) > 64 else ''}")
            for k in ks[:12]:
                # is it another element's value? search the row for a (code, scale) combination that matches
                v = eff[n0, k].item()
                cand = [(kk, bb) for kk in range(K) for bb in range(K // 64)
                        if abs(codes[n0, kk].item() * scales[n0, bb].item() - v) < 1e-3 * abs(v) + 1e-7]
                print(f"   k={k}: got {v:+.5f} want {Wd[n0, k].item():+.5f}; matches (code of k', scale of block b) for {cand[:6]}")

if __name__ == "__main__":
    main()
