"""CPU: the oracle (oracle/bnb4_oracle.c) reproduces every golden vector generated from the real
reference (tests/golden/make_golden.py). This is what pins the oracle on machines without
/root/reference (e.g. the GPU box)."""
import numpy as np
import pytest
import torch

from conftest import DT, QT, from_bits, golden, golden_8bit_maps, rel_err, same_values, same_values_ftz
from oracle import oracle as O

G = golden()
M8 = golden_8bit_maps()


def test_code_tables_bit_exact():
    for qt in ("nf4", "fp4"):
        assert np.array_equal(O.get_4bit_code(qt).view(torch.int32).numpy(), G[f"code/{qt}"]), qt


@pytest.mark.parametrize("i", range(int(G["q4/count"][0])))
def test_quantize_dequantize_4bit_vs_reference(i):
    qt_c, dt_c, bs, n = (int(v) for v in G[f"q4/{i}/meta"])
    A = from_bits(G[f"q4/{i}/A"], dt_c)
    packed, absmax = O.quantize_4bit(A, bs, QT[qt_c])
    assert np.array_equal(packed.reshape(-1).numpy(), G[f"q4/{i}/packed"]), "packed 4-bit codes differ"
    assert np.array_equal(absmax.view(torch.int32).numpy(), G[f"q4/{i}/absmax"]), "absmax differs"
    for oc, name in ((0, "fp32"), (2, "bf16"), (1, "fp16")):
        d = O.dequantize_4bit(packed, absmax, bs, QT[qt_c], (n,), DT[oc])
        cmp = same_values_ftz if name == "bf16" else same_values
        assert cmp(d, from_bits(G[f"q4/{i}/deq_{name}"], oc)), f"dequantize {name}"


@pytest.mark.parametrize("i", range(int(G["q8/count"][0])))
def test_blockwise_8bit_vs_reference(i):
    dt_c, bs, n = (int(v) for v in G[f"q8/{i}/meta"])
    A = from_bits(G[f"q8/{i}/A"], dt_c)
    code = from_bits(G["code/dynamic"], 0)
    for fma in (0, 1):  # both roundings of norm_to_lut_index agree with the reference binary
        q, am = O.quantize_blockwise(A, code, bs, fma)
        assert np.array_equal(q.numpy(), G[f"q8/{i}/q"])
        assert np.array_equal(am.view(torch.int32).numpy(), G[f"q8/{i}/absmax"])
    for oc, name in ((0, "fp32"), (2, "bf16"), (1, "fp16")):
        d = O.dequantize_blockwise(q, am, code, bs, DT[oc])
        assert same_values(d, from_bits(G[f"q8/{i}/deq_{name}"], oc))


@pytest.mark.parametrize("i", range(int(G["dq/count"][0])))
def test_double_quant_pieces_vs_reference(i):
    qt_c, dt_c, bs, N, K = (int(v) for v in G[f"dq/{i}/meta"])
    W = from_bits(G[f"dq/{i}/W"], dt_c)
    packed, absmax = O.quantize_4bit(W, bs, QT[qt_c])
    assert np.array_equal(packed.reshape(-1).numpy(), G[f"dq/{i}/packed"])
    # second level, with the reference's own offset (absmax.mean() is reduction-order dependent, SURVEY §8a note 7)
    offset = from_bits(G[f"dq/{i}/offset"], 0)
    code = from_bits(G["code/dynamic"], 0)
    q8, am2 = O.quantize_blockwise(absmax - offset, code, 256)
    assert np.array_equal(q8.numpy(), G[f"dq/{i}/absmax8"])
    assert np.array_equal(am2.view(torch.int32).numpy(), G[f"dq/{i}/absmax2"])
    am = O.dequantize_blockwise(q8, am2, code, 256, torch.float32) + offset
    d = O.dequantize_4bit(packed, am, bs, QT[qt_c], (N * K,), DT[dt_c])
    assert same_values(d, from_bits(G[f"dq/{i}/deq"], dt_c))


@pytest.mark.parametrize("i", range(int(G["gemm/count"][0])))
def test_gemm_4bit_vs_reference(i):
    qt_c, dt_c, bs, M, N, K, dqf, has_bias = (int(v) for v in G[f"gemm/{i}/meta"])
    x = from_bits(G[f"gemm/{i}/x"], dt_c).reshape(M, K)
    packed = torch.from_numpy(G[f"gemm/{i}/packed"]).reshape(-1, 1)
    bias = from_bits(G[f"gemm/{i}/bias"], dt_c) if has_bias else None
    kw = {}
    if dqf:
        absmax = from_bits(G[f"gemm/{i}/absmax2"], 0)
        kw = dict(absmax_8bit=torch.from_numpy(G[f"gemm/{i}/absmax8"]), absmax_code=from_bits(G["code/dynamic"], 0),
                  absmax_offset=from_bits(G[f"gemm/{i}/offset"], 0).reshape(()))
    else:
        absmax = from_bits(G[f"gemm/{i}/absmax"], 0)
    y, y32 = O.gemm_4bit(x, packed, (N, K), absmax, bs, QT[qt_c], bias, **kw)
    y_ref = from_bits(G[f"gemm/{i}/y"], dt_c).reshape(M, N)
    y32_ref = from_bits(G[f"gemm/{i}/y_fp32"], 0).reshape(M, N)
    # fp32-dequant + fp32-linear: only the summation order differs (double here, fp32 BLAS there)
    assert rel_err(y32, y32_ref) < 2e-6
    # reference CPU result (T-rounded weights, T output): one output ulp of T plus accumulation order
    tol = {0: 2e-6, 1: 1.5e-3, 2: 1e-2}[dt_c]
    assert rel_err(y, y_ref) < tol


@pytest.mark.parametrize("i", range(int(M8["m8/count"][0])))
def test_blockwise_8bit_other_code_maps_vs_reference(i):
    """Linear / fp8 / normal / few-bit / unsigned-dynamic code maps (zero-padded maps have repeated entries),
    incl. values on and one ulp either side of every decision boundary."""
    bs, n = (int(v) for v in M8[f"m8/{i}/meta"])
    code = from_bits(M8[f"m8/{i}/code"], 0)
    A = from_bits(M8[f"m8/{i}/A"], 0)
    q, am = O.quantize_blockwise(A, code, bs)
    name = str(M8[f"m8/{i}/name"])
    bad = np.nonzero(q.numpy() != M8[f"m8/{i}/q"])[0]
    assert bad.size == 0, f"{name} bs={bs}: {bad.size} codes differ, first x={A[int(bad[0])].item()!r}"
    assert np.array_equal(am.view(torch.int32).numpy(), M8[f"m8/{i}/absmax"]), name
    d = O.dequantize_blockwise(q, am, code, bs, torch.float32)
    assert same_values(d, from_bits(M8[f"m8/{i}/deq_fp32"], 0)), name
