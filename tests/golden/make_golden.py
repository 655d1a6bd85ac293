#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REAL reference
(bitsandbytes @ /root/reference, CPU backend, its own libbitsandbytes_cpu.so built by
oracle/build_ref.sh) on seeded inputs. Only runs where /root/reference exists; the resulting
``golden_4bit.npz`` is committed and is what travels to the GPU box.

    python tests/golden/make_golden.py

Every case stores the INPUT too (as raw uint16/float32 bits), so tests never depend on the RNG.
The reference has no stored golden vectors of its own for this path; its tests seed
everything to 0 (tests/conftest.py:9-14), which is what we do per case.

Cases (keys are ``<case>/<field>``):
  code/*          16-entry NF4/FP4 tables (functional.py:772-859) and the 256-entry dynamic map (:296-348)
  q4/<i>/*        quantize_4bit on the CPU backend (backends/default/ops.py:233-259): packed bytes + absmax,
                  and dequantize_4bit of them in fp32/bf16/fp16 (backends/cpu/ops.py:139-240 -> csrc/cpu_ops.cpp:304-434)
  q8/<i>/*        quantize_blockwise / dequantize_blockwise with the dynamic map (csrc/cpu_ops.cpp:436-665)
  dq/<i>/*        compress_statistics=True states (functional.py:938-951): uint8 absmax, state2.absmax, offset
  gemm/<i>/*      matmul_4bit on CPU (gemm_4bit default = dequantize_4bit + F.linear, default/ops.py:323-345)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.ref_import import import_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_4bit.npz")

_NP_VIEW = {torch.float32: (torch.int32, np.int32), torch.float16: (torch.int16, np.int16),
            torch.bfloat16: (torch.int16, np.int16)}
_DT_NAME = {torch.float32: "fp32", torch.float16: "fp16", torch.bfloat16: "bf16"}


def bits(t: torch.Tensor) -> np.ndarray:
    """Tensor -> numpy array of its raw bits (bf16 has no numpy dtype)."""
    t = t.detach().contiguous().cpu()
    if t.dtype in _NP_VIEW:
        return t.view(_NP_VIEW[t.dtype][0]).numpy().copy()
    return t.numpy().copy()


def make_input(n, dtype, seed, kind):
    g = torch.Generator().manual_seed(seed)
    if kind == "randn":
        x = torch.randn(n, generator=g)
    elif kind == "weight":  # W / sqrt(K)-like small values
        x = torch.randn(n, generator=g) * 0.02
    elif kind == "zeros_mixed":
        x = torch.randn(n, generator=g)
        x[: min(n, 130)] = 0.0  # whole zero blocks at the front
        x[::7] = 0.0
    elif kind == "ties":  # values sitting exactly on and next to code midpoints after scaling by absmax = 1
        x = torch.zeros(n)
        x[0] = 1.0
        vals = []
        for qt_code in (NF4, FP4):
            srt = torch.sort(qt_code).values
            mids = (srt[:-1] + srt[1:]) / 2
            for m in mids.tolist():
                m32 = np.float32(m)
                vals += [float(m32), float(np.nextafter(m32, np.float32(2))), float(np.nextafter(m32, np.float32(-2)))]
        vals += [0.0, -0.0, 1e-4, -1e-4, 0.0026, -0.0026, 0.00261, -0.00261, 1e-30, -1e-30]
        k = min(n - 1, len(vals))
        x[1 : 1 + k] = torch.tensor(vals[:k])
    elif kind == "denormal":
        x = torch.randn(n, generator=g) * 1e-39
    else:
        raise ValueError(kind)
    return x.to(dtype)


def main():
    global NF4, FP4
    bnb = import_reference()
    import bitsandbytes.functional as F

    store = {}
    NF4 = F.get_4bit_type("nf4", device="cpu")
    FP4 = F.get_4bit_type("fp4", device="cpu")
    dyn = F.create_dynamic_map().float()
    store["code/nf4"] = bits(NF4)
    store["code/fp4"] = bits(FP4)
    store["code/dynamic"] = bits(dyn)

    # ---------------------------------------------------------------- quantize_4bit / dequantize_4bit
    q4_cases = []
    for qt in ("nf4", "fp4"):
        for dt in (torch.float16, torch.bfloat16, torch.float32):
            q4_cases += [
                (qt, dt, 64, 64 * 96, "randn"),
                (qt, dt, 64, 64 * 5 + 17, "randn"),      # ragged tail (division path)
                (qt, dt, 64, 4097, "zeros_mixed"),       # odd n, zero blocks
                (qt, dt, 64, 256, "ties"),
            ]
        for bs in (32, 128, 256, 512, 1024, 2048, 4096):
            q4_cases += [(qt, torch.bfloat16, bs, bs * 3, "weight"), (qt, torch.float16, bs, bs * 2 + 35, "randn")]
        q4_cases += [(qt, torch.float32, 64, 1, "randn"), (qt, torch.float32, 64, 3, "randn"),
                     (qt, torch.float32, 64, 640, "denormal"), (qt, torch.float16, 64, 16384, "weight")]
    for i, (qt, dt, bs, n, kind) in enumerate(q4_cases):
        A = make_input(n, dt, 1000 + i, kind)
        packed, st = F.quantize_4bit(A, blocksize=bs, quant_type=qt)
        p = f"q4/{i}"
        store[f"{p}/meta"] = np.array([{"nf4": 2, "fp4": 1}[qt], {"fp32": 0, "fp16": 1, "bf16": 2}[_DT_NAME[dt]], bs, n])
        store[f"{p}/A"] = bits(A)
        store[f"{p}/packed"] = packed.numpy().reshape(-1).copy()
        store[f"{p}/absmax"] = bits(st.absmax)
        for odt in (torch.float32, torch.bfloat16, torch.float16):
            d = torch.ops.bitsandbytes.dequantize_4bit.default(packed, st.absmax, bs, qt, (n,), odt)
            store[f"{p}/deq_{_DT_NAME[odt]}"] = bits(d.reshape(-1))
    store["q4/count"] = np.array([len(q4_cases)])

    # ---------------------------------------------------------------- 8-bit blockwise (dynamic map)
    q8_cases = [(torch.float32, 256, 256 * 40, "randn"), (torch.float32, 256, 256 * 3 + 77, "zeros_mixed"),
                (torch.float32, 64, 64 * 9, "weight"), (torch.float32, 4096, 4096 * 2 + 5, "randn"),
                (torch.float16, 256, 2048, "randn"), (torch.bfloat16, 256, 2048, "randn")]
    for i, (dt, bs, n, kind) in enumerate(q8_cases):
        A = make_input(n, dt, 2000 + i, kind)
        if kind == "zeros_mixed":
            A[:256] = 0
        q, am = torch.ops.bitsandbytes.quantize_blockwise.default(A, dyn, bs)
        p = f"q8/{i}"
        store[f"{p}/meta"] = np.array([{"fp32": 0, "fp16": 1, "bf16": 2}[_DT_NAME[dt]], bs, n])
        store[f"{p}/A"] = bits(A)
        store[f"{p}/q"] = q.numpy().copy()
        store[f"{p}/absmax"] = bits(am)
        for odt in (torch.float32, torch.bfloat16, torch.float16):
            d = torch.ops.bitsandbytes.dequantize_blockwise.default(q, am, dyn, bs, odt)
            store[f"{p}/deq_{_DT_NAME[odt]}"] = bits(d)
    store["q8/count"] = np.array([len(q8_cases)])

    # ---------------------------------------------------------------- double quantisation states
    dq_cases = [("nf4", torch.bfloat16, 64, (64, 256)), ("fp4", torch.bfloat16, 128, (96, 512)),
                ("nf4", torch.float16, 64, (33, 128))]
    for i, (qt, dt, bs, shape) in enumerate(dq_cases):
        W = make_input(shape[0] * shape[1], dt, 3000 + i, "weight").reshape(shape)
        packed, st = F.quantize_4bit(W, blocksize=bs, quant_type=qt, compress_statistics=True)
        p = f"dq/{i}"
        store[f"{p}/meta"] = np.array([{"nf4": 2, "fp4": 1}[qt], {"fp32": 0, "fp16": 1, "bf16": 2}[_DT_NAME[dt]], bs,
                                      shape[0], shape[1]])
        store[f"{p}/W"] = bits(W)
        store[f"{p}/packed"] = packed.numpy().reshape(-1).copy()
        store[f"{p}/absmax8"] = st.absmax.numpy().copy()
        store[f"{p}/absmax2"] = bits(st.state2.absmax)
        store[f"{p}/offset"] = bits(st.offset.reshape(1))
        store[f"{p}/deq"] = bits(F.dequantize_4bit(packed, st).reshape(-1))
    store["dq/count"] = np.array([len(dq_cases)])

    # ---------------------------------------------------------------- gemm_4bit (matmul_4bit on CPU)
    gemm_cases = [
        # (qt, dtype, bs, M, N, K, double_quant, bias)
        ("nf4", torch.bfloat16, 64, 1, 64, 256, False, False),
        ("nf4", torch.bfloat16, 64, 1, 48, 2048, False, True),
        ("nf4", torch.float16, 64, 3, 40, 512, False, True),
        ("fp4", torch.bfloat16, 128, 1, 64, 512, True, False),
        ("nf4", torch.bfloat16, 64, 16, 64, 512, True, True),
        ("nf4", torch.float32, 64, 2, 32, 256, False, False),
        ("fp4", torch.float16, 64, 33, 48, 256, False, True),
        ("nf4", torch.bfloat16, 32, 1, 32, 128, False, False),
        # MFMA-sized: tall tiles and enough K for cross-workgroup K slices (reach the producer/consumer kernel and the
        # multi-row-tile register-transposed kernel with reference-generated vectors)
        ("nf4", torch.bfloat16, 64, 64, 256, 4096, False, True),
        ("nf4", torch.bfloat16, 64, 64, 384, 2048, True, True),
        ("fp4", torch.float16, 128, 16, 128, 4096, True, False),
    ]
    for i, (qt, dt, bs, M, N, K, dqf, use_bias) in enumerate(gemm_cases):
        W = (make_input(N * K, torch.float32, 4000 + i, "randn").reshape(N, K) / (K**0.5)).to(dt)
        x = make_input(M * K, torch.float32, 4100 + i, "randn").reshape(M, K).to(dt)
        bias = make_input(N, torch.float32, 4200 + i, "randn").to(dt) if use_bias else None
        packed, st = F.quantize_4bit(W, blocksize=bs, quant_type=qt, compress_statistics=dqf)
        y = bnb.matmul_4bit(x, packed, st, bias=bias)
        # fp32 tolerance oracle of SURVEY §8a note 8: fp32 dequant + fp32 linear
        st32_absmax = st.absmax if not dqf else (F.dequantize_blockwise(st.absmax, st.state2) + st.offset)
        W32 = torch.ops.bitsandbytes.dequantize_4bit.default(packed, st32_absmax.float(), bs, qt, (N, K), torch.float32)
        y32 = torch.nn.functional.linear(x.float(), W32.reshape(N, K), None if bias is None else bias.float())
        p = f"gemm/{i}"
        store[f"{p}/meta"] = np.array([{"nf4": 2, "fp4": 1}[qt], {"fp32": 0, "fp16": 1, "bf16": 2}[_DT_NAME[dt]], bs, M, N,
                                      K, int(dqf), int(use_bias)])
        store[f"{p}/x"] = bits(x)
        if N * K <= 262144:  # (the MFMA-sized cases do not store their dense weight: no test reads it)
            store[f"{p}/W"] = bits(W)
        if bias is not None:
            store[f"{p}/bias"] = bits(bias)
        store[f"{p}/packed"] = packed.numpy().reshape(-1).copy()
        if dqf:
            store[f"{p}/absmax8"] = st.absmax.numpy().copy()
            store[f"{p}/absmax2"] = bits(st.state2.absmax)
            store[f"{p}/offset"] = bits(st.offset.reshape(1))
        else:
            store[f"{p}/absmax"] = bits(st.absmax)
        store[f"{p}/y"] = bits(y)
        store[f"{p}/y_fp32"] = bits(y32)
    store["gemm/count"] = np.array([len(gemm_cases)])

    np.savez_compressed(OUT, **store)
    print(f"wrote {OUT}: {len(store)} arrays, {os.path.getsize(OUT) / 1024:.1f} KiB, reference {bnb.__version__}")


if __name__ == "__main__":
    main()
