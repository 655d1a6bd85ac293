"""CPU-side check of the 8-bit encoder ALGORITHM of csrc/blockwise8.hip (threshold bins, 1024-cell table, dense-cell
search), re-stated in numpy and compared with the oracle on every discretisation bin for every code-map
constructor - including few-bit maps whose zero padding puts up to 253 thresholds into one cell. A development aid
(run without a GPU); the on-device parity test is tests/test_gpu_parity.py::test_blockwise_8bit_other_code_maps."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bitsandbytes_amd.functional as F  # noqa: E402
from oracle import oracle as O  # noqa: E402

f32 = np.float32
def bin_value(u):
    return f32(-1.0) + (f32(2.0) * np.asarray(u, dtype=f32)) / f32(65535.0)

def build(code):
    code = code.astype(f32)
    mid = np.full(256, np.inf, dtype=f32)
    mid[:255] = f32(0.5) * (code[:255] + code[1:])
    thr = np.zeros(256, dtype=np.int64)
    for t in range(256):
        lo, hi = 0, 65536
        for _ in range(17):
            if lo >= hi: break
            c = (lo + hi) >> 1
            if bin_value(c) > mid[t]: hi = c
            else: lo = c + 1
        thr[t] = hi
    below_s = np.zeros(1025, dtype=np.int64)
    for c in range(1025):
        first = c * 64
        below = 0
        step = 128
        while step >= 1:
            if below + step - 1 < 255 and thr[below + step - 1] < first:
                below += step
            step >>= 1
        below_s[c] = below & 0xFF  # uint8 store
    cell = np.zeros(1024, dtype=np.int64)
    for c in range(1024):
        below = below_s[c]
        inside = int(below_s[c + 1]) - int(below)
        off = (thr[below] - c * 64) if inside > 0 else 0
        cell[c] = (below | ((inside & 0xFFFFFFFF) << 8) | (off << 16)) & 0xFFFFFFFF
    return thr, cell

def encode(x, inv, thr, cell):
    v = np.clip(f32(x) * f32(inv), f32(-1), f32(1)).astype(f32)
    t = ((v + f32(1)) * f32(0.5)).astype(f32)
    p = (t * f32(65535.0)).astype(f32)
    r = (p + f32(0.5)).astype(f32)
    u = r.astype(np.int64) & 0xFFFF
    out = np.zeros(len(x), dtype=np.uint8)
    for i, ui in enumerate(u):
        ce = int(cell[ui >> 6])
        q = ce & 0xFF
        cnt = (ce >> 8) & 0xFF
        if cnt > 1:
            lo = 0
            step = 128
            while step >= 1:
                if lo + step - 1 < cnt and thr[q + lo + step - 1] <= ui:
                    lo += step
                step >>= 1
            q += lo
        else:
            q += 1 if (cnt == 1 and (ui & 63) >= (ce >> 16)) else 0
        out[i] = q & 0xFF
    return out

codes = {
 "dynamic": F.create_dynamic_map(), "dynamic_unsigned": F.create_dynamic_map(signed=False),
 "linear8": F.create_linear_map(True, 8), "linear4": F.create_linear_map(True, 4), "linear2": F.create_linear_map(True, 2),
 "linear8u": F.create_linear_map(False, 8),
 "fp8_e4m3": F.create_fp8_map(True, 4, 3, 8), "fp8_e5m2": F.create_fp8_map(True, 5, 2, 8),
 "fp4_as_map": F.create_fp8_map(True, 2, 1, 4), "normal": F.create_normal_map(),
 "dyn3": F.create_dynamic_map(True, 3, 3),
}
g = torch.Generator().manual_seed(3)
for name, code in codes.items():
    thr, cell = build(code.numpy())
    # all 65536 bins with absmax 1 (block of 256: first element 1.0) + random data
    u = torch.arange(65536, dtype=torch.float64)
    centres = (-1.0 + 2.0 * u / 65535.0)
    vals = torch.cat([centres, centres + 0.499 / 65535.0, centres - 0.499 / 65535.0]).clamp(-1, 1).float()
    if "unsigned" in name or name.endswith("u"):
        pass
    pad = (-vals.numel()) % 255
    vals = torch.cat([vals, torch.zeros(pad)])
    A = torch.cat([torch.ones(vals.numel() // 255, 1), vals.view(-1, 255)], dim=1).reshape(-1).contiguous()
    q_o, am_o = O.quantize_blockwise(A, code, 256)
    q_e = encode(A.numpy(), 1.0, thr, cell)
    bad = np.nonzero(q_e != q_o.numpy())[0]
    print(f"{name:18s} thresholds max-per-cell={max((c>>8)&0xFF for c in cell):3d}  mismatches={len(bad)}", ("first x=%r emu=%d oracle=%d" % (A[bad[0]].item(), q_e[bad[0]], q_o[bad[0]])) if len(bad) else "")
