"""Exhaustive check of the 4-bit encoder: every fp32 bit pattern with |x| <= 1 (2 * 0x3F800001 values)
is quantized on the GPU and by the CPU oracle, codes compared bit for bit.

Blocks of 4096 start with the element 1.0, so absmax = 1 and the scaled value is the pattern itself.
Takes a few minutes (the oracle runs ~15 ns/element on one host core). Run once per encoder change:

    python tests/checks/exhaustive_quantize.py [--quant-type nf4|fp4|both] [--stride 1]
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bitsandbytes_amd.functional as F  # noqa: E402
from oracle import oracle as O  # noqa: E402

BS = 4096
SLICE = (BS - 1) * 8192  # patterns per slice


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quant-type", default="both")
    ap.add_argument("--stride", type=int, default=1)
    a = ap.parse_args()
    qts = ("nf4", "fp4") if a.quant_type == "both" else (a.quant_type,)
    top = 0x3F800000  # 1.0f
    for qt in qts:
        t0 = time.time()
        checked = 0
        bad_total = 0
        for sign in (0, 0x80000000):
            start = 0
            while start <= top:
                cnt = min(SLICE * a.stride, top + 1 - start)
                pat = np.arange(start, start + cnt, a.stride, dtype=np.int64)
                k = len(pat)
                pad = (-k) % (BS - 1)
                pat = np.concatenate([pat, np.zeros(pad, dtype=np.int64)])
                vals = (pat.astype(np.uint32) | np.uint32(sign)).view(np.float32).reshape(-1, BS - 1)
                A = np.concatenate([np.ones((vals.shape[0], 1), dtype=np.float32), vals], axis=1).reshape(-1)
                At = torch.from_numpy(A)
                q_o, am_o = O.quantize_4bit(At, BS, qt)
                q, st = F.quantize_4bit(At.cuda(), blocksize=BS, quant_type=qt)
                assert torch.equal(st.absmax.cpu(), am_o)
                nbad = int((q.cpu() != q_o).sum())
                if nbad:
                    idx = int((q.cpu() != q_o).nonzero()[0])
                    print(f"{qt}: MISMATCH sign={sign:#x} start={start:#x}: {nbad} bytes, first byte {idx} "
                          f"values {A[2 * idx]!r} {A[2 * idx + 1]!r} gpu={int(q.view(-1)[idx]):#x} oracle={int(q_o.view(-1)[idx]):#x}")
                bad_total += nbad
                checked += k
                start += cnt
        print(f"{qt}: {checked} fp32 patterns in [-1, 1] checked (stride {a.stride}), {bad_total} mismatching bytes, "
              f"{time.time() - t0:.0f} s")
    return 0


if __name__ == "__main__":
    sys.exit(main())
