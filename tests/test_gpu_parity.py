"""GPU (-m gpu): parity of the HIP path against the oracle and the committed golden vectors, through
the public Python API and straight through the C ABI. Bars (BASELINE.json north_star):
  * quantize_4bit packed codes + absmax, 8-bit codes: BIT-EXACT
  * dequantize: bit-exact (up to the sign of zero / bf16 denormal flush noted in conftest.py)
  * fused dequant+matmul: relative Frobenius error <= 1e-2 vs fp32-dequantize + fp32-linear
    (REL_TOL below; observed values are ~1e-3 for bf16, ~2e-4 for fp16)
"""
import os
import ctypes as ct
import warnings

import numpy as np
import pytest
import torch

from conftest import DT, QT, from_bits, golden, golden_8bit_maps, gpu_ready, rel_err, same_values_ftz
from oracle import oracle as O

pytestmark = pytest.mark.gpu

REL_TOL = 1e-2  # north_star tolerance for the bf16/fp16 dequant+matmul
DEV = "cuda"
G = golden()
M8 = golden_8bit_maps()


@pytest.fixture(scope="module", autouse=True)
def _require_gpu_and_native_lib():
    # A host without any AMD GPU device node (a plain `pytest tests` in a CPU container): skip. A GPU box whose torch
    # cannot see the device, or whose native library did not load, must FAIL - never pass on a fallback.
    import os

    if not gpu_ready() and not os.path.exists("/dev/kfd") and os.environ.get("BNB_REQUIRE_GPU") != "1":
        pytest.skip("no GPU device on this host (set BNB_REQUIRE_GPU=1 to make this an error)", allow_module_level=False)
    assert gpu_ready(), "GPU tests selected but torch.cuda.is_available() is False"
    import bitsandbytes_amd as bnb

    assert bnb.lib, "libbitsandbytes_mi355x.so is not loaded: GPU tests must run on the native HIP path"
    yield


def _F():
    import bitsandbytes_amd.functional as F

    return F


# ------------------------------------------------------------------------------------------ quantize / dequantize
@pytest.mark.parametrize("i", range(int(G["q4/count"][0])))
def test_quantize_dequantize_4bit_golden(i):
    F = _F()
    qt_c, dt_c, bs, n = (int(v) for v in G[f"q4/{i}/meta"])
    A = from_bits(G[f"q4/{i}/A"], dt_c).to(DEV)
    packed, st = F.quantize_4bit(A, blocksize=bs, quant_type=QT[qt_c])
    assert np.array_equal(packed.cpu().reshape(-1).numpy(), G[f"q4/{i}/packed"]), "packed codes differ from reference"
    assert np.array_equal(st.absmax.cpu().view(torch.int32).numpy(), G[f"q4/{i}/absmax"]), "absmax differs"
    for oc, name in ((0, "fp32"), (2, "bf16"), (1, "fp16")):
        d = torch.ops.bitsandbytes.dequantize_4bit.default(packed, st.absmax, bs, QT[qt_c], (n,), DT[oc])
        assert same_values_ftz(d.cpu(), from_bits(G[f"q4/{i}/deq_{name}"], oc)), f"dequantize -> {name}"


@pytest.mark.parametrize("quant_type", ["nf4", "fp4"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32], ids=["fp16", "bf16", "fp32"])
@pytest.mark.parametrize("blocksize", [32, 64, 128, 256, 512, 1024, 2048, 4096])
def test_quantize_4bit_random_vs_oracle(quant_type, dtype, blocksize):
    F = _F()
    for n in (blocksize * 40 + 13, 8192 * 3 + 1, 7):
        A = (torch.randn(n) * 0.3).to(dtype)
        A[:: 11] = 0
        if n > 4 * blocksize:
            A[blocksize : 2 * blocksize] = 0  # a whole zero block
        q_o, am_o = O.quantize_4bit(A, blocksize, quant_type)
        q, st = F.quantize_4bit(A.to(DEV), blocksize=blocksize, quant_type=quant_type)
        assert torch.equal(q.cpu(), q_o), f"n={n}"
        assert torch.equal(st.absmax.cpu(), am_o), f"n={n}"
        d = F.dequantize_4bit(q, st)
        assert same_values_ftz(d.cpu(), O.dequantize_4bit(q_o, am_o, blocksize, quant_type, A.shape, dtype))


@pytest.mark.parametrize("quant_type", ["nf4", "fp4"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32], ids=["fp16", "bf16", "fp32"])
@pytest.mark.parametrize("blocksize", [32, 64, 512, 2048, 4096])
def test_quantize_4bit_wide_tiles(quant_type, dtype, blocksize):
    """n large enough for the 4-chunk (8192-element) workgroup tiles, with a ragged tail."""
    F = _F()
    n = 4 * 256 * 8192 + 2048 * 3 + 5
    A = (torch.randn(n) * 0.3).to(dtype)
    A[::13] = 0
    q_o, am_o = O.quantize_4bit(A, blocksize, quant_type)
    q, st = F.quantize_4bit(A.to(DEV), blocksize=blocksize, quant_type=quant_type)
    assert torch.equal(q.cpu(), q_o)
    assert torch.equal(st.absmax.cpu(), am_o)


def _bound_windows(quant_type, half_width):
    """fp32 values in (-1, 1]: every ulp within +-half_width ulps of each decision bound and of a sample
    of the kernel's cell edges, plus exact code values, +-0 and denormals."""
    import numpy as np

    code = F_code(quant_type)
    srt = np.sort(code.astype(np.float32))
    bounds = ((srt[:-1] + srt[1:]) / np.float32(2)).astype(np.float32)
    S = 16 if quant_type == "nf4" else 512
    edges = ((np.arange(0, 2 * S + 1, dtype=np.float64) - S + 0.5) / S).astype(np.float32)
    edges = edges[np.abs(edges) < 1][:: max(1, len(edges) // 64)]
    centres = np.concatenate([bounds, edges, srt, np.float32([0.0, 1e-40, -1e-40, 1.17549435e-38])])
    offs = np.arange(-half_width, half_width + 1, dtype=np.int64)
    out = []
    for c in centres:
        bits = np.float32(c).view(np.int32).astype(np.int64)
        # walk ulps in sign-magnitude space
        key = np.where(bits < 0, -(bits & 0x7FFFFFFF), bits) + offs
        b = np.where(key < 0, (-key) | 0x80000000, key).astype(np.uint32)
        out.append(b.view(np.float32))
    v = np.concatenate(out + [np.float32([0.0, -0.0])])
    v = v[np.isfinite(v) & (np.abs(v) <= 1)]
    return torch.from_numpy(v.copy())


def F_code(quant_type):
    return O.get_4bit_code(quant_type).numpy()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "fp32"])
@pytest.mark.parametrize("blocksize", [64, 4096])
def test_quantize_4bit_fp4_pipelined_form_vs_oracle(dtype, blocksize):
    """FP4 tensors of whole aligned tiles from 16 M elements up take the grid-strided, prefetching form of quantize4_kernel
    (csrc/quantize4.hip, PIPE): codes and absmax bit-exact against the oracle at that size, and identical to the one-tile form
    (tuning knob)."""
    import bitsandbytes_amd as bnb
    from oracle import oracle as O

    F = _F()
    g = torch.Generator().manual_seed(21)
    n = 4096 * 4096 + (4096 * 64 if blocksize == 64 else 0)
    A = (torch.randn(n, generator=g) * 0.3).to(dtype)
    A[::4099] = 0
    q, st = F.quantize_4bit(A.to(DEV), blocksize=blocksize, quant_type="fp4")
    q_o, am_o = O.quantize_4bit(A, blocksize, "fp4")
    assert torch.equal(q.cpu().flatten(), q_o.flatten()) and torch.equal(st.absmax.cpu(), am_o)
    try:
        bnb.lib.bnb_mi355x_set_tuning(3, 0, 0, 0)
        q1, st1 = F.quantize_4bit(A.to(DEV), blocksize=blocksize, quant_type="fp4")
    finally:
        bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
    assert torch.equal(q1, q) and torch.equal(st1.absmax, st.absmax)


@pytest.mark.parametrize("quant_type", ["nf4", "fp4"])
def test_quantize_4bit_every_ulp_around_bounds(quant_type):
    """The cell-table encoder must agree with the 15-bound count on every float next to a decision bound or
    a cell edge. Blocks are pinned to absmax = 1 (first element 1.0) so that s = x exactly."""
    F = _F()
    bs = 64
    v = _bound_windows(quant_type, 2048)
    pad = (-len(v)) % (bs - 1)
    v = torch.cat([v, torch.zeros(pad)])
    A = torch.cat([torch.ones(len(v) // (bs - 1), 1), v.view(-1, bs - 1)], dim=1).reshape(-1).contiguous()
    # scaled copies exercise x * (1/absmax) with a non-trivial reciprocal
    for scale in (1.0, 3.0, 0.37, 1e-30, 6e4):
        As = (A * scale).float()
        q_o, am_o = O.quantize_4bit(As, bs, quant_type)
        q, st = F.quantize_4bit(As.to(DEV), blocksize=bs, quant_type=quant_type)
        assert torch.equal(st.absmax.cpu(), am_o)
        bad = (q.cpu() != q_o).nonzero()
        assert bad.numel() == 0, f"scale={scale}: {bad.numel()} bytes differ, first at {bad[0].item()}"


@pytest.mark.parametrize("quant_type", ["nf4", "fp4"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32], ids=["fp16", "bf16", "fp32"])
def test_quantize_4bit_inf_nan_blocks(quant_type, dtype):
    """inf / NaN inputs follow the torch semantics of the reference CPU backend: NaN anywhere in a block ->
    absmax NaN and every code of the block = the last sorted position; inf -> absmax inf, inf*0 = NaN."""
    F = _F()
    for bs, n in ((64, 64 * 70 + 9), (256, 256 * 40), (4096, 4096 * 3 + 100)):
        A = (torch.randn(n) * 0.5).to(dtype)
        A[5] = float("inf")
        A[bs + 3] = float("-inf")
        A[2 * bs + 7] = float("nan")
        A[3 * bs] = float("nan")
        A[3 * bs + 1] = float("inf")
        A[n - 2] = float("nan")  # in the tail block when there is one
        q_o, am_o = O.quantize_4bit(A, bs, quant_type)
        q, st = F.quantize_4bit(A.to(DEV), blocksize=bs, quant_type=quant_type)
        am = st.absmax.cpu()
        assert torch.equal(torch.isnan(am), torch.isnan(am_o))
        assert torch.equal(torch.nan_to_num(am, nan=-1.0), torch.nan_to_num(am_o, nan=-1.0))
        assert torch.equal(q.cpu(), q_o), f"bs={bs}"


def test_config1_full_size_4096x4096_fp16_nf4():
    """BASELINE config 1 at full size: every one of the 16.7M codes and 262144 absmax bit-exact."""
    F = _F()
    W = torch.randn(4096, 4096).half()
    q_o, am_o = O.quantize_4bit(W, 64, "nf4")
    q, st = F.quantize_4bit(W.to(DEV), blocksize=64, quant_type="nf4")
    assert q.shape == (4096 * 4096 // 2, 1) and torch.equal(q.cpu(), q_o) and torch.equal(st.absmax.cpu(), am_o)
    d = F.dequantize_4bit(q, st)
    assert d.shape == W.shape and d.dtype == torch.float16
    assert torch.equal(d.cpu(), O.dequantize_4bit(q_o, am_o, 64, "nf4", W.shape, torch.float16))
    # reference envelope (tests/test_functional.py:606-651): NF4 bs64 mean-abs 0.072798 +- 7 sigma
    assert (d.float().cpu() - W.float()).abs().mean().item() < 0.072798 + 7 * 0.000074 + 2e-4
    # idempotence: re-quantising the dequantised weight reproduces the same codes
    q2, st2 = F.quantize_4bit(d, blocksize=64, quant_type="nf4")
    assert torch.equal(q2, q)


@pytest.mark.parametrize("quant_type", ["nf4", "fp4"])
def test_quantize_4bit_maximum_size(quant_type):
    """The reference's size limit (tests/test_functional.py:698-716, `test_4bit_quant_large`): 2**31 - 1 elements,
    the largest count a C `int` carries. Odd, with a 63-element ragged last block. Slices from the start, from
    across the 2**30 / 2**31-byte offsets and from the ragged end are checked bit for bit against the oracle."""
    F = _F()
    n = 2**31 - 1
    free, _ = torch.cuda.mem_get_info()
    if free < 14 * 2**30:
        pytest.skip("needs ~11 GiB of device memory")
    A = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    step = 2**27
    g = torch.Generator(device=DEV).manual_seed(5)
    for s0 in range(0, n, step):  # filled in pieces: a 2**31-element fp32 randn temp is not needed
        A[s0 : s0 + step].normal_(generator=g)
    q, st = F.quantize_4bit(A, blocksize=64, quant_type=quant_type)
    assert q.dtype == torch.uint8 and q.numel() == 2**30 and st.absmax.numel() == 2**25
    d = F.dequantize_4bit(q, st)
    assert d.shape == (n,) and d.dtype == torch.bfloat16
    span = 64 * 1000
    starts = [0, 2**29 - span // 2, 2**30 - span // 2, 2**30 + 2**29, n - 63 - span]
    for s0 in starts:
        assert s0 % 64 == 0
        e0 = n if s0 == starts[-1] else s0 + span  # whole blocks, except the ragged end of the tensor
        a = A[s0:e0].cpu()
        q_o, am_o = O.quantize_4bit(a, 64, quant_type)
        assert torch.equal(q.reshape(-1)[s0 // 2 : (e0 + 1) // 2].cpu(), q_o.reshape(-1)), f"codes at {s0}"
        assert torch.equal(st.absmax[s0 // 64 : (e0 + 63) // 64].cpu(), am_o), f"absmax at {s0}"
        assert same_values_ftz(d[s0:e0].cpu(), O.dequantize_4bit(q_o, am_o, 64, quant_type, a.shape, torch.bfloat16))
    del A, d
    with pytest.raises(ValueError, match="2\\*\\*31"):
        torch.ops.bitsandbytes.dequantize_4bit.default(q, st.absmax, 64, quant_type, (2**31,), torch.bfloat16)


@pytest.mark.parametrize("storage", [torch.bfloat16, torch.float16, torch.float32])
def test_quant_storage_views(storage):
    F = _F()
    W = torch.randn(64, 256, device=DEV).bfloat16()
    q8, st8 = F.quantize_4bit(W, quant_type="nf4")
    q, st = F.quantize_4bit(W, quant_type="nf4", quant_storage=storage)
    assert q.dtype == storage and torch.equal(q.view(torch.uint8).reshape(-1), q8.reshape(-1))
    assert torch.equal(F.dequantize_4bit(q, st), F.dequantize_4bit(q8, st8))


def test_noncontiguous_and_offset_inputs():
    F = _F()
    base = torch.randn(130, 257, device=DEV).bfloat16()
    A = base[1:, 1:]  # non-contiguous, and (once made contiguous by the op) arbitrary values
    q, st = F.quantize_4bit(A, quant_type="nf4")
    q_o, am_o = O.quantize_4bit(A.cpu().contiguous(), 64, "nf4")
    assert torch.equal(q.cpu(), q_o) and torch.equal(st.absmax.cpu(), am_o)
    # mis-aligned input pointer (storage offset of 1 element) exercises the scalar-load path
    flat = torch.randn(64 * 50 + 1, device=DEV).half()
    B = flat[1:]
    q, st = F.quantize_4bit(B, quant_type="fp4")
    q_o, am_o = O.quantize_4bit(B.cpu(), 64, "fp4")
    assert torch.equal(q.cpu(), q_o) and torch.equal(st.absmax.cpu(), am_o)
    out = torch.empty(64 * 50 + 8, device=DEV, dtype=torch.half)[3 : 3 + 64 * 50]  # mis-aligned output
    F.dequantize_4bit(q, st, out=out)
    assert same_values_ftz(out.cpu(), O.dequantize_4bit(q_o, am_o, 64, "fp4", (64 * 50,), torch.half))


# ------------------------------------------------------------------------------------------ 8-bit + double quant
@pytest.mark.parametrize("i", range(int(G["q8/count"][0])))
def test_blockwise_8bit_golden(i):
    dt_c, bs, n = (int(v) for v in G[f"q8/{i}/meta"])
    A = from_bits(G[f"q8/{i}/A"], dt_c).to(DEV)
    code = from_bits(G["code/dynamic"], 0).to(DEV)
    q, am = torch.ops.bitsandbytes.quantize_blockwise.default(A, code, bs)
    assert np.array_equal(q.cpu().numpy(), G[f"q8/{i}/q"])
    assert np.array_equal(am.cpu().view(torch.int32).numpy(), G[f"q8/{i}/absmax"])
    for oc, name in ((0, "fp32"), (2, "bf16"), (1, "fp16")):
        d = torch.ops.bitsandbytes.dequantize_blockwise.default(q, am, code, bs, DT[oc])
        assert same_values_ftz(d.cpu(), from_bits(G[f"q8/{i}/deq_{name}"], oc))


def test_blockwise_8bit_dense_vs_oracle():
    """Dense sweep near zero, where the dynamic map is densest and the 65536-bin rule differs from
    nearest-code (SURVEY §8a note 6)."""
    F = _F()
    code = F.create_dynamic_map()
    A = torch.cat([torch.linspace(-1, 1, 100003), torch.linspace(-2e-3, 2e-3, 50021), torch.randn(4096) * 1e-5])
    A[0] = 1.0
    pad = (-A.numel()) % 256
    A = torch.cat([A, torch.zeros(pad)])
    A.view(-1, 256)[:, 0] = 1.0  # absmax == 1 in every block -> x/absmax == x exactly
    q_o, am_o = O.quantize_blockwise(A, code, 256)
    q, am = torch.ops.bitsandbytes.quantize_blockwise.default(A.to(DEV), code.to(DEV), 256)
    assert torch.equal(q.cpu(), q_o) and torch.equal(am.cpu(), am_o)


@pytest.mark.parametrize("quant_type,blocksize", [("nf4", 64), ("fp4", 128), ("nf4", 32), ("nf4", 4096)])
def test_dequantize_4bit_round5_tile_forms_vs_oracle(quant_type, blocksize):
    """Round 5 gave dequantize4 size-dependent launch forms (csrc/dequantize4.hip): fp32 outputs of whole 4096-element tiles take the
    LINE-CONTIGUOUS kernel (one 16-byte store per unit), 16-bit outputs below 2^25 elements 8 packed dwords per lane, from there on 4.
    Every form against the oracle, the fast paths against the forms they replaced (tuning knob 10 + v) bit for bit, and the sizes that
    must fall back (a ragged tail, an unaligned output view) too."""
    import bitsandbytes_amd as bnb

    F = _F()
    for n in (4096, 4096 * 24, 4096 * 24 + blocksize, 1 << 20, (1 << 25) + 4096 * 3):
        if n > (1 << 22) and (quant_type, blocksize) != ("nf4", 64):
            continue  # (the large sizes once: the oracle is scalar C)
        g = torch.Generator().manual_seed(n % 1000)
        A = (torch.randn(n, generator=g) * 0.3)
        A[::13] = 0
        q_o, am_o = O.quantize_4bit(A, blocksize, quant_type)
        q, am = q_o.to(DEV), am_o.to(DEV)
        for dt in (torch.float32, torch.bfloat16, torch.float16):
            want = O.dequantize_4bit(q_o, am_o, blocksize, quant_type, A.shape, dt)
            d = torch.ops.bitsandbytes.dequantize_4bit.default(q, am, blocksize, quant_type, (n,), dt)
            assert same_values_ftz(d.cpu(), want), (n, dt)
            for knob in (14, 18, 40) if dt != torch.float32 else (40, 32, 38):  # forced general u = 4 / 8, round 4's form; fp32: general, lines u = 2 / 8
                try:
                    bnb.lib.bnb_mi355x_set_tuning(knob, 0, 0, 0)
                    d2 = torch.ops.bitsandbytes.dequantize_4bit.default(q, am, blocksize, quant_type, (n,), dt)
                finally:
                    bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
                assert torch.equal(d2.view(torch.int32 if dt == torch.float32 else torch.int16), d.view(torch.int32 if dt == torch.float32 else torch.int16)), (n, dt, knob)
    # an output view that is only 4-byte aligned: the fp32 fast path must decline (16-byte stores), same values
    n = 4096 * 8
    A = torch.randn(n) * 0.3
    q_o, am_o = O.quantize_4bit(A, blocksize, quant_type)
    buf = torch.empty(n + 1, device=DEV)
    out = buf[1:]
    torch.ops.bitsandbytes.dequantize_4bit.out(q_o.to(DEV), am_o.to(DEV), blocksize, quant_type, (n,), torch.float32, out=out)
    assert same_values_ftz(out.cpu(), O.dequantize_4bit(q_o, am_o, blocksize, quant_type, A.shape, torch.float32))


def test_dequantize_8bit_four_units_per_lane_equals_the_first_form():
    """dequantize8_kernel keeps four 4-element units in flight per lane since round 5 (one before: 51 % of the HBM peak). Sizes
    around its 4096-element iteration - whole iterations, a tail that takes the scalar branch unit by unit, fewer elements than
    one iteration - against the oracle and against the first form (tuning knob 6), every output dtype."""
    import bitsandbytes_amd as bnb

    F = _F()
    code = F.create_dynamic_map()
    for n in (4096 * 5, 4096 * 5 + 1024 + 3, 4096 + 2, 1000, 3, 256 * 1024 + 777):
        A = torch.randn(n) * 0.2
        q_o, am_o = O.quantize_blockwise(A, code, 256)
        q, am = q_o.to(DEV), am_o.to(DEV)
        for dt in (torch.float32, torch.bfloat16, torch.float16):
            d = torch.ops.bitsandbytes.dequantize_blockwise.default(q, am, code.to(DEV), 256, dt)
            assert same_values_ftz(d.cpu(), O.dequantize_blockwise(q_o, am_o, code, 256, dt)), (n, dt)
            try:
                bnb.lib.bnb_mi355x_set_tuning(6, 0, 0, 0)
                d1 = torch.ops.bitsandbytes.dequantize_blockwise.default(q, am, code.to(DEV), 256, dt)
            finally:
                bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
            assert torch.equal(d1.view(torch.int32 if dt == torch.float32 else torch.int16), d.view(torch.int32 if dt == torch.float32 else torch.int16)), (n, dt)


@pytest.mark.parametrize("blocksize", [64, 128, 256, 512, 1024, 2048, 4096])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16], ids=["fp32", "fp16", "bf16"])
def test_blockwise_8bit_general_vs_oracle(blocksize, dtype):
    """8-bit blockwise as an op in its own right: every blocksize and input dtype, ragged sizes,
    all-zero blocks, unaligned views; codes and absmax bit-exact, dequantize bit-exact."""
    F = _F()
    code = F.create_dynamic_map()
    for n in (blocksize * 37 + 5, 3, 256 * 1024 + 777):
        A = (torch.randn(n) * 0.1).to(dtype)
        A[::7] = 0
        if n > 3 * blocksize:
            A[blocksize : 2 * blocksize] = 0  # an all-zero block -> codes 0, absmax 0
        q_o, am_o = O.quantize_blockwise(A, code, blocksize)
        q, am = torch.ops.bitsandbytes.quantize_blockwise.default(A.to(DEV), code.to(DEV), blocksize)
        assert torch.equal(q.cpu(), q_o), f"n={n}"
        assert torch.equal(am.cpu(), am_o), f"n={n}"
        for out_dtype in (torch.float32, dtype):
            d = torch.ops.bitsandbytes.dequantize_blockwise.default(q, am, code.to(DEV), blocksize, out_dtype)
            assert same_values_ftz(d.cpu(), O.dequantize_blockwise(q_o, am_o, code, blocksize, out_dtype))
    # unaligned view (offset by one element): scalar path
    base = (torch.randn(blocksize * 9 + 1) * 0.3).to(dtype).to(DEV)
    view = base[1:]
    q_o, am_o = O.quantize_blockwise(view.cpu().contiguous(), code, blocksize)
    q, am = torch.ops.bitsandbytes.quantize_blockwise.default(view, code.to(DEV), blocksize)
    assert torch.equal(q.cpu(), q_o) and torch.equal(am.cpu(), am_o)


@pytest.mark.parametrize("which", ["dynamic_unsigned", "linear8", "linear4", "linear2", "fp8_e4m3", "fp8_e5m2",
                                   "fp4_as_map", "normal", "dynamic3"])
def test_blockwise_8bit_other_code_maps(which):
    """Code maps other than the signed dynamic one: the reference's linear / fp8 / normal / few-bit constructors
    (few-bit maps are zero-padded to 256 entries, so up to 253 decision thresholds fall into ONE cell of the
    encoder's table - the dense-cell search path). Every discretisation bin +- just under half a bin with absmax
    pinned to 1, plus random blocks; codes, absmax and the dequantized values bit-exact against the oracle."""
    F = _F()
    code = {
        "dynamic_unsigned": lambda: F.create_dynamic_map(signed=False),
        "linear8": lambda: F.create_linear_map(True, 8),
        "linear4": lambda: F.create_linear_map(True, 4),
        "linear2": lambda: F.create_linear_map(True, 2),
        "fp8_e4m3": lambda: F.create_fp8_map(True, 4, 3, 8),
        "fp8_e5m2": lambda: F.create_fp8_map(True, 5, 2, 8),
        "fp4_as_map": lambda: F.create_fp8_map(True, 2, 1, 4),
        "normal": lambda: F.create_normal_map(),
        "dynamic3": lambda: F.create_dynamic_map(True, 3, 3),
    }[which]()
    u = torch.arange(65536, dtype=torch.float64)
    centres = -1.0 + 2.0 * u / 65535.0
    vals = torch.cat([centres, centres + 0.499 / 65535.0, centres - 0.499 / 65535.0]).clamp(-1, 1).float()
    vals = torch.cat([vals, torch.zeros((-vals.numel()) % 255)])
    A = torch.cat([torch.ones(vals.numel() // 255, 1), vals.view(-1, 255)], dim=1).reshape(-1).contiguous()
    R = torch.randn(256 * 61 + 17) * 0.2
    R[256:512] = 0
    for data, bs in ((A, 256), (R, 256), (R, 64), (R, 2048)):
        q_o, am_o = O.quantize_blockwise(data, code, bs)
        q, am = torch.ops.bitsandbytes.quantize_blockwise.default(data.to(DEV), code.to(DEV), bs)
        bad = (q.cpu() != q_o).nonzero()
        assert bad.numel() == 0, f"bs={bs}: {bad.numel()} codes differ, first x={data[bad[0]].item()!r}"
        assert torch.equal(am.cpu(), am_o)
        d = torch.ops.bitsandbytes.dequantize_blockwise.default(q, am, code.to(DEV), bs, torch.float32)
        assert same_values_ftz(d.cpu(), O.dequantize_blockwise(q_o, am_o, code, bs, torch.float32))


@pytest.mark.parametrize("i", range(int(M8["m8/count"][0])))
def test_blockwise_8bit_other_code_maps_golden(i):
    """The same code maps against vectors produced by the reference itself (tests/golden/golden_8bit_maps.npz)."""
    bs, n = (int(v) for v in M8[f"m8/{i}/meta"])
    code = from_bits(M8[f"m8/{i}/code"], 0).to(DEV)
    A = from_bits(M8[f"m8/{i}/A"], 0).to(DEV)
    name = str(M8[f"m8/{i}/name"])
    q, am = torch.ops.bitsandbytes.quantize_blockwise.default(A, code, bs)
    bad = np.nonzero(q.cpu().numpy() != M8[f"m8/{i}/q"])[0]
    assert bad.size == 0, f"{name} bs={bs}: {bad.size} codes differ, first x={A[int(bad[0])].item()!r}"
    assert np.array_equal(am.cpu().view(torch.int32).numpy(), M8[f"m8/{i}/absmax"]), name
    d = torch.ops.bitsandbytes.dequantize_blockwise.default(q, am, code, bs, torch.float32)
    assert same_values_ftz(d.cpu(), from_bits(M8[f"m8/{i}/deq_fp32"], 0)), name


@pytest.mark.parametrize("variant", [1, 2], ids=["cell-table", "byte-table"])
def test_blockwise_8bit_both_encoders(variant):
    """The 8-bit quantize has two encoders chosen by input size (cell table below 2**20 elements, the reference's own
    65536-entry byte table - held in LDS - above). Forced one at a time: every discretisation bin +- just under half a bin,
    every blocksize and dtype with ragged sizes and all-zero blocks, and the code maps whose thresholds crowd one cell."""
    import bitsandbytes_amd as bnb

    F = _F()
    try:
        bnb.lib.bnb_mi355x_set_tuning(variant, 0, 0, 0)
        code = F.create_dynamic_map()
        u = torch.arange(65536, dtype=torch.float64)
        centres = -1.0 + 2.0 * u / 65535.0
        vals = torch.cat([centres, centres + 0.499 / 65535.0, centres - 0.499 / 65535.0, centres + 0.501 / 65535.0]).clamp(-1, 1).float()
        vals = torch.cat([vals, torch.zeros((-vals.numel()) % 255)])
        A = torch.cat([torch.ones(vals.numel() // 255, 1), vals.view(-1, 255)], dim=1).reshape(-1).contiguous()
        q_o, am_o = O.quantize_blockwise(A, code, 256)
        q, am = torch.ops.bitsandbytes.quantize_blockwise.default(A.to(DEV), code.to(DEV), 256)
        assert torch.equal(q.cpu(), q_o) and torch.equal(am.cpu(), am_o)
        for blocksize in (64, 256, 1024, 2048, 4096):
            for dtype in (torch.float32, torch.float16, torch.bfloat16):
                for n in (blocksize * 37 + 5, 3, 70001):
                    X = (torch.randn(n) * 0.1).to(dtype)
                    X[::7] = 0
                    if n > 3 * blocksize:
                        X[blocksize: 2 * blocksize] = 0
                    q_o, am_o = O.quantize_blockwise(X, code, blocksize)
                    q, am = torch.ops.bitsandbytes.quantize_blockwise.default(X.to(DEV), code.to(DEV), blocksize)
                    assert torch.equal(q.cpu(), q_o) and torch.equal(am.cpu(), am_o), (blocksize, dtype, n)
        R = torch.randn(256 * 61 + 17) * 0.2
        for mk in (lambda: F.create_linear_map(True, 2), lambda: F.create_fp8_map(True, 2, 1, 4), lambda: F.create_normal_map(),
                   lambda: F.create_dynamic_map(signed=False)):
            c2 = mk()
            for data, bs in ((A, 256), (R, 64)):
                q_o, am_o = O.quantize_blockwise(data, c2, bs)
                q, am = torch.ops.bitsandbytes.quantize_blockwise.default(data.to(DEV), c2.to(DEV), bs)
                assert torch.equal(q.cpu(), q_o) and torch.equal(am.cpu(), am_o)
        # unaligned view: scalar loads / byte stores
        base = (torch.randn(256 * 9 + 1) * 0.3).to(DEV)
        q_o, am_o = O.quantize_blockwise(base[1:].cpu().contiguous(), code, 256)
        q, am = torch.ops.bitsandbytes.quantize_blockwise.default(base[1:], code.to(DEV), 256)
        assert torch.equal(q.cpu(), q_o) and torch.equal(am.cpu(), am_o)
    finally:
        bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)


def test_blockwise_8bit_large_input_takes_the_byte_table_kernel():
    """3 M elements through the default dispatch (byte-table kernel, persistent workgroups with a prefetched unit), ragged
    tail, bit-exact against the oracle; dequantize round trip within the map's resolution."""
    F = _F()
    code = F.create_dynamic_map()
    n = 3 * 1024 * 1024 + 333
    A = torch.randn(n) * 0.05
    A[5000:5256] = 0
    q_o, am_o = O.quantize_blockwise(A, code, 256)
    q, am = torch.ops.bitsandbytes.quantize_blockwise.default(A.to(DEV), code.to(DEV), 256)
    assert torch.equal(q.cpu(), q_o) and torch.equal(am.cpu(), am_o)
    d = torch.ops.bitsandbytes.dequantize_blockwise.default(q, am, code.to(DEV), 256, torch.float32)
    assert same_values_ftz(d.cpu(), O.dequantize_blockwise(q_o, am_o, code, 256, torch.float32))


def test_blockwise_8bit_every_bin():
    """All 65536 discretisation bins (and their neighbourhood) with absmax pinned to 1: the threshold /
    cell-table encoder must reproduce the reference's 64K-entry table exactly."""
    F = _F()
    code = F.create_dynamic_map()
    u = torch.arange(65536, dtype=torch.float64)
    centres = (-1.0 + 2.0 * u / 65535.0)
    vals = torch.cat([centres, centres + 1.0 / 65535.0 * 0.499, centres - 1.0 / 65535.0 * 0.499,
                      centres + 1.0 / 65535.0 * 0.501]).clamp(-1, 1).float()
    pad = (-vals.numel()) % 255
    vals = torch.cat([vals, torch.zeros(pad)])
    A = torch.cat([torch.ones(vals.numel() // 255, 1), vals.view(-1, 255)], dim=1).reshape(-1).contiguous()
    q_o, am_o = O.quantize_blockwise(A, code, 256)
    q, am = torch.ops.bitsandbytes.quantize_blockwise.default(A.to(DEV), code.to(DEV), 256)
    bad = (q.cpu() != q_o).nonzero()
    assert bad.numel() == 0, f"{bad.numel()} codes differ, first at {bad[0].item()}: x={A[bad[0]].item()!r}"
    assert torch.equal(am.cpu(), am_o)


@pytest.mark.parametrize("quant_type,blocksize", [("nf4", 64), ("fp4", 128)])
def test_double_quant_state_vs_oracle(quant_type, blocksize):
    F = _F()
    W = (torch.randn(512, 1024) * 0.02).bfloat16()
    q, st = F.quantize_4bit(W.to(DEV), blocksize=blocksize, quant_type=quant_type, compress_statistics=True)
    assert st.nested and st.absmax.dtype == torch.uint8 and st.state2.blocksize == 256
    q_o, am_o = O.quantize_4bit(W, blocksize, quant_type)
    assert torch.equal(q.cpu(), q_o)
    # offset = absmax.mean() is a device-order reduction (SURVEY §8a note 7): take the device's own
    # offset, then everything downstream must be bit-exact
    offset = st.offset.cpu()
    assert abs(offset.item() - am_o.mean().item()) <= 2e-7 * abs(am_o.mean().item()) + 1e-12
    q8_o, am2_o = O.quantize_blockwise(am_o - offset, F.create_dynamic_map(), 256)
    assert torch.equal(st.absmax.cpu(), q8_o) and torch.equal(st.state2.absmax.cpu(), am2_o)
    d = F.dequantize_4bit(q, st)
    am_rec = O.dequantize_blockwise(q8_o, am2_o, F.create_dynamic_map(), 256, torch.float32) + offset
    assert same_values_ftz(d.cpu(), O.dequantize_4bit(q_o, am_rec, blocksize, quant_type, W.shape, torch.bfloat16))
    # reference envelope test_4bit_compressed_stats (tests/test_functional.py:666-695)
    err = (d.float().cpu() - W.float()).abs().mean() / W.float().abs().mean()
    assert err < 0.28


def _tree_mean_f32(a: np.ndarray) -> np.float32:
    """csrc/blockwise8.hip's summation order for the mean of the fp32 absmax, restated: steps of 1024 elements as a balanced binary
    tree in index order (zeros past the end); the steps of a chunk (ceil(n / (1024 * 256)) of them) one after the other; the <= 256
    chunk sums as one more balanced tree of 256 leaves; one IEEE division by n."""
    a = np.asarray(a, dtype=np.float32)
    n = a.size
    steps = -(-n // (1024 * 256))
    chunks = -(-n // (1024 * steps))
    x = np.zeros(chunks * steps * 1024, dtype=np.float32)
    x[:n] = a
    t = x.reshape(chunks * steps, 1024)
    while t.shape[1] > 1:
        t = t[:, 0::2] + t[:, 1::2]
    t = t.reshape(chunks, steps)
    part = t[:, 0].copy()
    for s_ in range(1, steps):
        part = part + t[:, s_]
    p = np.zeros(256, dtype=np.float32)
    p[:chunks] = part
    while p.size > 1:
        p = p[0::2] + p[1::2]
    return np.float32(p[0]) / np.float32(n)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("quant_type,blocksize,n", [
    ("nf4", 64, 512 * 1024), ("fp4", 128, 4096 * 4096), ("nf4", 64, 4096 * 4096), ("nf4", 32, 1000 * 96 + 32), ("fp4", 64, 64 * 255 + 7),
    ("nf4", 64, 64 * 1025), ("nf4", 4096, 4096 * 3 + 1), ("nf4", 32, 32 * (1024 * 256 + 1029)), ("nf4", 64, 5)])
def test_quantize_4bit_nested_one_call_equals_the_four_operator_sequence(dtype, quant_type, blocksize, n):
    """quantize_4bit(compress_statistics=True) runs as ONE operator (three launches). Packed bytes = the plain operator's; the offset
    = the mean of the fp32 absmax in the documented summation order, bit for bit (numpy restatement above) and within 2 ulp-ish of
    the float64 mean; 8-bit codes and second-level absmax = quantize_blockwise(absmax - offset, 256), operator AND oracle, bit for bit."""
    F = _F()
    torch.manual_seed(n % 9973)
    W = (torch.randn(n) * 0.03).to(dtype)
    W[::97] *= 11.0
    Wd = W.to(DEV)
    q, st = F.quantize_4bit(Wd, blocksize=blocksize, quant_type=quant_type, compress_statistics=True)
    q_plain, am_plain = torch.ops.bitsandbytes.quantize_4bit.default(Wd, blocksize, quant_type, torch.uint8)
    assert torch.equal(q, q_plain)
    assert st.nested and st.absmax.dtype == torch.uint8 and st.absmax.shape == am_plain.shape and st.state2.blocksize == 256
    assert st.offset.shape == () and st.offset.dtype == torch.float32 and st.state2.code.dtype == torch.float32
    am = am_plain.cpu()
    off = st.offset.cpu()
    want = _tree_mean_f32(am.numpy())
    assert off.numpy().view(np.int32) == np.float32(want).view(np.int32), (off.item(), float(want))
    exact = float(am.double().mean())
    assert abs(off.item() - exact) <= 3e-7 * abs(exact) + 1e-12
    code = F.create_dynamic_map()
    q8, am2 = torch.ops.bitsandbytes.quantize_blockwise.default((am - off).to(DEV), code.to(DEV), 256)
    assert torch.equal(st.absmax, q8) and torch.equal(st.state2.absmax, am2)
    if am.numel() <= 1 << 20:
        q8_o, am2_o = O.quantize_blockwise(am - off, code, 256)
        assert torch.equal(st.absmax.cpu(), q8_o) and torch.equal(st.state2.absmax.cpu(), am2_o)
    assert torch.equal(st.state2.code.cpu(), code)
    # twice the same bits (fixed order, no atomics), also from another launch geometry's neighbourhood: a view at an odd offset
    q_b, st_b = F.quantize_4bit(Wd, blocksize=blocksize, quant_type=quant_type, compress_statistics=True)
    assert torch.equal(q_b, q) and torch.equal(st_b.absmax, st.absmax) and torch.equal(st_b.offset.view(torch.int32), st.offset.view(torch.int32))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("quant_type,blocksize,shape", [
    ("nf4", 64, (4096, 4096)), ("fp4", 128, (1024, 1024)), ("nf4", 32, (1000, 96)), ("nf4", 64, (64 * 255 + 7,)), ("fp4", 4096, (3, 4096 + 512)),
    ("nf4", 64, (5,)), ("nf4", 64, (8192, 4096 + 64))])
def test_dequantize_4bit_nested_one_launch_equals_the_three_operator_sequence(dtype, quant_type, blocksize, shape):
    """dequantize_4bit of a double-quantised state runs as ONE operator / launch (the scale of every block reconstructed in the kernel:
    code2[q] * absmax2 + offset, two roundings). Bit for bit the reference's sequence dequantize_blockwise -> += offset ->
    dequantize_4bit through the plain operators (each checked against the oracle elsewhere), and against the oracle itself where the
    size allows. Both tile sizes of the kernel (below / from 2^25 elements), ragged ends, every dtype."""
    F = _F()
    torch.manual_seed(blocksize + len(shape))
    W = (torch.randn(*shape) * 0.05).to(dtype).to(DEV)
    q, st = F.quantize_4bit(W, blocksize=blocksize, quant_type=quant_type, compress_statistics=True)
    got = F.dequantize_4bit(q, st)
    assert got.shape == W.shape and got.dtype == dtype
    am = torch.ops.bitsandbytes.dequantize_blockwise.default(st.absmax, st.state2.absmax, st.state2.code, 256, torch.float32)
    am = am + st.offset
    want = torch.ops.bitsandbytes.dequantize_4bit.default(q, am, blocksize, quant_type, list(W.shape), dtype)
    assert torch.equal(got.view(torch.int16 if dtype != torch.float32 else torch.int32), want.view(torch.int16 if dtype != torch.float32 else torch.int32))
    if W.numel() <= 1 << 21:
        code = st.state2.code.cpu()
        am_o = O.dequantize_blockwise(st.absmax.cpu(), st.state2.absmax.cpu(), code, 256, torch.float32) + st.offset.cpu()
        assert torch.equal(am_o, am.cpu())
        assert same_values_ftz(got.cpu(), O.dequantize_4bit(q.cpu(), am_o, blocksize, quant_type, W.shape, dtype))
    # the out= form keeps the three-operator sequence: same bits
    out = torch.empty_like(W)
    F.dequantize_4bit(q, st, out=out)
    assert torch.equal(out.view(got.dtype), got)


def test_unfused_gemm_with_nested_statistics_equals_the_plain_statistics_call():
    """M above the fused range with double-quantised statistics: one dequantize launch (statistics reconstructed in it) + the library
    GEMM. The same call with the reconstructed fp32 absmax handed over must give the same bits (same weights, same GEMM)."""
    F = _F()
    from bitsandbytes_amd.backends import hip

    torch.manual_seed(5)
    W = (torch.randn(1024, 2048) * 0.03).bfloat16().to(DEV)
    q, st = F.quantize_4bit(W, blocksize=64, quant_type="nf4", compress_statistics=True)
    x = torch.randn(640, 2048, device=DEV).bfloat16()
    am = torch.ops.bitsandbytes.dequantize_blockwise.default(st.absmax, st.state2.absmax, st.state2.code, 256, torch.float32) + st.offset
    a = hip._gemm_4bit_unfused(x, q, st.shape, st.state2.absmax, 64, "nf4", None, st.absmax, st.state2.code, st.offset)
    b = hip._gemm_4bit_unfused(x, q, st.shape, am, 64, "nf4", None, None, None, None)
    assert torch.equal(a, b)
    # a state rebuilt by QuantState.from_dict under a 16-bit default dtype (model loaders set one) carries a 16-bit offset: the host
    # sequence promotes it in `absmax + offset`; the one-launch form must take the same value
    off16 = st.offset.to(torch.float16)
    am16 = torch.ops.bitsandbytes.dequantize_blockwise.default(st.absmax, st.state2.absmax, st.state2.code, 256, torch.float32) + off16
    a16 = hip._gemm_4bit_unfused(x, q, st.shape, st.state2.absmax, 64, "nf4", None, st.absmax, st.state2.code, off16)
    b16 = hip._gemm_4bit_unfused(x, q, st.shape, am16, 64, "nf4", None, None, None, None)
    assert am16.dtype == torch.float32 and torch.equal(a16, b16)
    y = torch.ops.bitsandbytes.gemm_4bit.default(x, q, st.shape, st.state2.absmax, 64, "nf4", absmax_8bit=st.absmax,
                                                 absmax_code=st.state2.code, absmax_offset=st.offset)
    assert rel_err(y, x.float() @ F.dequantize_4bit(q, st).float().t()) < REL_TOL


# ------------------------------------------------------------------------------------------ gemm / gemv
def _oracle_y(x, q, st, bias=None):
    """fp32-dequant + fp32-linear oracle result for a (possibly nested) QuantState living on the GPU."""
    kw = {}
    if st.nested:
        absmax = st.state2.absmax
        kw = dict(absmax_8bit=st.absmax, absmax_code=st.state2.code, absmax_offset=st.offset)
    else:
        absmax = st.absmax
    return O.gemm_4bit(x, q, st.shape, absmax, st.blocksize, st.quant_type, bias, **kw)[1]


def _run_kernel(kernel, x, q, st, bias=None):
    """Call the fused op with an explicit kernel choice (0 auto, 1 / 3 streaming dot kernel, 2 MFMA kernels)."""
    from bitsandbytes_amd.backends import hip

    if st.nested:
        args = (x, q, st.shape, st.state2.absmax, st.blocksize, st.quant_type, bias, st.absmax, st.state2.code, st.offset)
    else:
        args = (x, q, st.shape, st.absmax, st.blocksize, st.quant_type, bias, None, None, None)
    return hip._gemm_4bit_fused(*args, kernel=kernel)


@pytest.mark.parametrize("i", range(int(G["gemm/count"][0])))
def test_gemm_4bit_golden(i):
    import bitsandbytes_amd as bnb

    F = _F()
    qt_c, dt_c, bs, M, N, K, dqf, has_bias = (int(v) for v in G[f"gemm/{i}/meta"])
    x = from_bits(G[f"gemm/{i}/x"], dt_c).reshape(M, K).to(DEV)
    packed = torch.from_numpy(G[f"gemm/{i}/packed"]).reshape(-1, 1).to(DEV)
    bias = from_bits(G[f"gemm/{i}/bias"], dt_c).to(DEV) if has_bias else None
    code = F.get_4bit_type(QT[qt_c], device=DEV)
    if dqf:
        s2 = F.QuantState(absmax=from_bits(G[f"gemm/{i}/absmax2"], 0).to(DEV), code=from_bits(G["code/dynamic"], 0).to(DEV),
                          blocksize=256, dtype=torch.float32)
        st = F.QuantState(absmax=torch.from_numpy(G[f"gemm/{i}/absmax8"]).to(DEV), shape=torch.Size((N, K)), code=code,
                          blocksize=bs, quant_type=QT[qt_c], dtype=DT[dt_c],
                          offset=from_bits(G[f"gemm/{i}/offset"], 0).reshape(()).to(DEV), state2=s2)
    else:
        st = F.QuantState(absmax=from_bits(G[f"gemm/{i}/absmax"], 0).to(DEV), shape=torch.Size((N, K)), code=code,
                          blocksize=bs, quant_type=QT[qt_c], dtype=DT[dt_c])
    y = bnb.matmul_4bit(x, packed, st, bias=bias)
    y32_ref = from_bits(G[f"gemm/{i}/y_fp32"], 0).reshape(M, N)
    y_ref = from_bits(G[f"gemm/{i}/y"], dt_c).reshape(M, N)
    tol = 2e-5 if dt_c == 0 else REL_TOL
    assert y.shape == (M, N) and y.dtype == DT[dt_c]
    assert rel_err(y.cpu(), y32_ref) < tol
    # and never worse than ~2x the reference CPU backend's own distance to the fp32 result
    if dt_c != 0:
        assert rel_err(y.cpu(), y32_ref) < max(2.5 * rel_err(y_ref, y32_ref), 3e-3)


@pytest.mark.parametrize("i", [8, 9, 10])
@pytest.mark.parametrize("knob", [0, 1104, 1402, 1204, 1302, 2000, 2002, 2100])
def test_gemm_4bit_golden_mfma_geometries(i, knob):
    """The MFMA-sized reference-generated vectors (tests/golden/make_golden.py: M = 64 / 16, K up to 4096) through every MFMA
    kernel family and cross-workgroup K slices: producer/consumer (cfg 11 / 12 / 13 / 14), register-transposed (cfg 20 / 21),
    and the production routing (knob 0)."""
    import bitsandbytes_amd as bnb

    F = _F()
    qt_c, dt_c, bs, M, N, K, dqf, has_bias = (int(v) for v in G[f"gemm/{i}/meta"])
    x = from_bits(G[f"gemm/{i}/x"], dt_c).reshape(M, K).to(DEV)
    packed = torch.from_numpy(G[f"gemm/{i}/packed"]).reshape(-1, 1).to(DEV)
    bias = from_bits(G[f"gemm/{i}/bias"], dt_c).to(DEV) if has_bias else None
    code = F.get_4bit_type(QT[qt_c], device=DEV)
    if dqf:
        s2 = F.QuantState(absmax=from_bits(G[f"gemm/{i}/absmax2"], 0).to(DEV), code=from_bits(G["code/dynamic"], 0).to(DEV),
                          blocksize=256, dtype=torch.float32)
        st = F.QuantState(absmax=torch.from_numpy(G[f"gemm/{i}/absmax8"]).to(DEV), shape=torch.Size((N, K)), code=code,
                          blocksize=bs, quant_type=QT[qt_c], dtype=DT[dt_c],
                          offset=from_bits(G[f"gemm/{i}/offset"], 0).reshape(()).to(DEV), state2=s2)
    else:
        st = F.QuantState(absmax=from_bits(G[f"gemm/{i}/absmax"], 0).to(DEV), shape=torch.Size((N, K)), code=code,
                          blocksize=bs, quant_type=QT[qt_c], dtype=DT[dt_c])
    try:
        bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, knob)
        y = _run_kernel(2 if knob else 0, x, packed, st, bias)
    finally:
        bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
    y32_ref = from_bits(G[f"gemm/{i}/y_fp32"], 0).reshape(M, N)
    y_ref = from_bits(G[f"gemm/{i}/y"], dt_c).reshape(M, N)
    assert rel_err(y.cpu(), y32_ref) < REL_TOL
    assert rel_err(y.cpu(), y32_ref) < max(2.5 * rel_err(y_ref, y32_ref), 3e-3)


SHAPES = [  # (M, N, K)
    (1, 4096, 4096),   # BASELINE config 2
    (1, 1000, 2048), (1, 31, 96), (1, 512, 11008 // 4), (2, 256, 4096), (3, 130, 1024), (4, 4096, 1024),
    (5, 256, 1024), (8, 4096, 4096), (16, 512, 2048), (17, 96, 512), (33, 200, 768), (64, 1024, 4096),
    (65, 128, 512), (130, 64, 256),
    (200, 384, 1024), (256, 1024, 2048), (300, 130, 768), (512, 512, 1024),   # tall batches: several row passes of the MFMA kernels
]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
def test_gemm_4bit_shapes_both_kernels(M, N, K, dtype):
    F = _F()
    W = (torch.randn(N, K) / K**0.5).to(dtype)
    x = torch.randn(M, K).to(dtype)
    bias = torch.randn(N).to(dtype)
    q, st = F.quantize_4bit(W.to(DEV), blocksize=64, quant_type="nf4")
    y_ref = _oracle_y(x, q, st, bias)
    for kernel in (1, 2, 0):
        if kernel == 2 and K % 256:
            continue
        if kernel == 1 and M > 130:
            continue  # (the streaming kernel is exact at any M, and slow there: covered up to 130 rows)
        y = _run_kernel(kernel, x.to(DEV), q, st, bias.to(DEV))
        e = rel_err(y.cpu(), y_ref)
        assert e < REL_TOL, f"kernel={kernel} rel err {e}"


@pytest.mark.parametrize("M,N,K", [(1, 256, 8192), (2, 256, 11008), (2, 130, 8192), (4, 128, 8192), (3, 64, 14336),
                                   (1, 64, 32768), (1, 16, 53248), (2, 24, 26624), (7, 40, 6144), (1, 300, 2048),
                                   (2, 300, 2048), (4, 300, 1024)])
@pytest.mark.parametrize("dq", [False, True], ids=["absmax32", "nested"])
def test_dot_kernel_long_and_short_rows(M, N, K, dq):
    """The dot kernel's activation image in LDS: several loop iterations over K, images that only fit with
    fewer activation rows per pass (K = 14336 at M = 3), rows too long for any image (K = 53248 falls back to
    per-wavefront loads) and the single-segment geometry (K <= 2048)."""
    F = _F()
    W = (torch.randn(N, K) / K**0.5).bfloat16()
    x = torch.randn(M, K).bfloat16()
    bias = torch.randn(N).bfloat16()
    q, st = F.quantize_4bit(W.to(DEV), blocksize=64, quant_type="nf4", compress_statistics=dq)
    y_ref = _oracle_y(x, q, st, bias)
    y = _run_kernel(1, x.to(DEV), q, st, bias.to(DEV))
    assert rel_err(y.cpu(), y_ref) < REL_TOL
    assert torch.equal(y, _run_kernel(1, x.to(DEV), q, st, bias.to(DEV)))  # bit-reproducible


@pytest.mark.parametrize("quant_type,blocksize,dq", [("fp4", 128, True), ("nf4", 64, True), ("fp4", 64, False),
                                                      ("nf4", 32, False), ("nf4", 256, True), ("nf4", 4096, False)])
@pytest.mark.parametrize("M", [1, 4, 16, 48])
def test_gemm_4bit_quant_variants(quant_type, blocksize, dq, M):
    """Covers BASELINE config 5 (FP4, double quant, blocksize 128, bf16, N = K = 4096 at M = 1)."""
    import bitsandbytes_amd as bnb

    F = _F()
    N, K = (4096, 4096) if M == 1 else (512, 4096)
    W = (torch.randn(N, K) / K**0.5).bfloat16()
    x = torch.randn(M, K).bfloat16()
    q, st = F.quantize_4bit(W.to(DEV), blocksize=blocksize, quant_type=quant_type, compress_statistics=dq)
    y = bnb.matmul_4bit(x.to(DEV), q, st)
    assert rel_err(y.cpu(), _oracle_y(x, q, st)) < REL_TOL
    if blocksize >= 64:
        y2 = _run_kernel(2, x.to(DEV), q, st)
        assert rel_err(y2.cpu(), _oracle_y(x, q, st)) < REL_TOL


def test_config3_m64_8192_subset_and_properties():
    """BASELINE config 3 (M = 64, N = K = 8192) at full size: oracle on a row subset (the oracle is
    scalar C), plus size-independent properties: linearity in x, and exact column independence."""
    import bitsandbytes_amd as bnb

    F = _F()
    N = K = 8192
    M = 64
    W = (torch.randn(N, K, device=DEV) / K**0.5).bfloat16()
    x = torch.randn(M, K, device=DEV).bfloat16()
    q, st = F.quantize_4bit(W, blocksize=64, quant_type="nf4")
    del W
    y = bnb.matmul_4bit(x, q, st)
    assert y.shape == (M, N)
    rows = torch.cat([torch.arange(0, 16), torch.arange(4000, 4016), torch.arange(8176, 8192)])
    bpr, blk = K // 2, K // 64
    q_sub = torch.cat([q.view(N, bpr)[r] for r in rows]).cpu().reshape(-1, 1)
    am_sub = torch.cat([st.absmax.view(N, blk)[r] for r in rows]).cpu()
    y_sub = O.gemm_4bit(x.cpu(), q_sub, (len(rows), K), am_sub, 64, "nf4")[1]
    assert rel_err(y[:, rows].cpu(), y_sub) < REL_TOL
    # linearity: (x1 + x2) W == x1 W + x2 W up to bf16 rounding of inputs/outputs
    x2 = torch.randn(M, K, device=DEV).bfloat16()
    lhs = bnb.matmul_4bit((x.float() + x2.float()).bfloat16(), q, st).float()
    rhs = y.float() + bnb.matmul_4bit(x2, q, st).float()
    assert rel_err(lhs.cpu(), rhs.cpu()) < 2e-2
    # determinism
    assert torch.equal(bnb.matmul_4bit(x, q, st), y)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32], ids=["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("dq", [False, True])
def test_gemv_4bit_legacy_api_and_eye(dtype, dq):
    """F.gemv_4bit (reference functional.py:1300-1334) incl. the identity-weight check of the reference's
    test_gemv_eye_4bit (tests/test_functional.py:950-977): 0 and 1 are exact NF4 codes, so y == x."""
    F = _F()
    dim = 256
    eye = torch.eye(dim, device=DEV, dtype=dtype)
    q, st = F.quantize_4bit(eye, quant_type="nf4", compress_statistics=dq)
    x = torch.randn(1, 1, dim, device=DEV, dtype=dtype)
    y = F.gemv_4bit(x, q, state=st)
    tol = dict(rtol=1e-2, atol=1e-2) if dq else dict(rtol=0, atol=0)  # double quant perturbs absmax=1 slightly
    torch.testing.assert_close(y, x, **tol)
    W = (torch.randn(384, 512) / 512**0.5).to(dtype)
    q, st = F.quantize_4bit(W.to(DEV), quant_type="fp4", compress_statistics=dq)
    x = torch.randn(1, 512).to(dtype)
    y = F.gemv_4bit(x.to(DEV), q, state=st)
    assert rel_err(y.cpu(), _oracle_y(x, q, st)) < (2e-5 if dtype == torch.float32 else REL_TOL)


def test_gemm_fp32_and_generic_fallbacks():
    """fp32 activations (dot kernel's generic path for M <= 4, dequant + rocBLAS above), odd K,
    K % blocksize != 0 (warns, unfused path like the reference backends/cuda/ops.py:956-962)."""
    import bitsandbytes_amd as bnb

    F = _F()
    for (M, N, K) in [(1, 64, 256), (3, 100, 512), (9, 64, 256)]:
        W = torch.randn(N, K) / K**0.5
        x = torch.randn(M, K)
        q, st = F.quantize_4bit(W.to(DEV), quant_type="nf4")
        assert rel_err(bnb.matmul_4bit(x.to(DEV), q, st).cpu(), _oracle_y(x, q, st)) < 2e-5
    W = (torch.randn(48, 96) / 10).bfloat16()  # K = 96 is not a multiple of blocksize 64
    x = torch.randn(2, 96).bfloat16()
    q, st = F.quantize_4bit(W.to(DEV), quant_type="nf4")
    with pytest.warns(UserWarning, match="not aligned"):
        y = bnb.matmul_4bit(x.to(DEV), q, st)
    assert rel_err(y.cpu(), _oracle_y(x, q, st)) < REL_TOL
    # the C-ABI kernels themselves also accept it (generic kernel): K odd, flat block indexing
    W = (torch.randn(10, 33) / 5).half()
    x = torch.randn(1, 33).half()
    q, st = F.quantize_4bit(W.to(DEV), quant_type="nf4", blocksize=32)
    y = _run_kernel(1, x.to(DEV), q, st)
    assert rel_err(y.cpu(), _oracle_y(x, q, st)) < REL_TOL


# ------------------------------------------------------------------------------------------ C ABI, graphs, modules
def test_c_abi_direct_calls():
    """Drive the reference-named symbols with raw pointers, exactly as the reference's ctypes layer
    does (backends/cuda/ops.py:384-420, 846-901)."""
    from bitsandbytes_amd.cextension import lib

    N, K, bs = 256, 1024, 64
    W = (torch.randn(N, K) / K**0.5).bfloat16()
    Wd = W.to(DEV)
    absmax = torch.empty(N * K // bs, device=DEV, dtype=torch.float32)
    packed = torch.empty(N * K // 2, device=DEV, dtype=torch.uint8)
    torch.cuda.synchronize()
    lib.cquantize_blockwise_bf16_nf4(None, Wd.data_ptr(), absmax.data_ptr(), packed.data_ptr(), bs, N * K)  # NULL stream
    torch.cuda.synchronize()
    q_o, am_o = O.quantize_4bit(W, bs, "nf4")
    assert torch.equal(packed.cpu(), q_o.reshape(-1)) and torch.equal(absmax.cpu(), am_o)
    stream = torch._C._cuda_getCurrentRawStream(0)
    out = torch.empty(N, K, device=DEV, dtype=torch.bfloat16)
    lib.cdequantize_blockwise_bf16_nf4(None, packed.data_ptr(), absmax.data_ptr(), out.data_ptr(), bs, N * K, stream)
    assert same_values_ftz(out.cpu(), O.dequantize_4bit(q_o, am_o, bs, "nf4", (N, K), torch.bfloat16))
    for M in (1, 16):
        x = torch.randn(M, K).bfloat16()
        xd = x.to(DEV)
        y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        lib.cgemm_4bit_bf16(xd.data_ptr(), packed.data_ptr(), absmax.data_ptr(), None, None, None, y.data_ptr(), None,
                            M, N, K, bs, 2, stream)
        assert rel_err(y.cpu(), O.gemm_4bit(x, q_o, (N, K), am_o, bs, "nf4")[1]) < REL_TOL
    code = O.get_4bit_code("nf4").to(DEV)
    y = torch.empty(1, N, device=DEV, dtype=torch.bfloat16)
    lib.cgemm_4bit_inference_naive_bf16(N, 1, K, xd[:1].contiguous().data_ptr(), packed.data_ptr(), absmax.data_ptr(),
                                        code.data_ptr(), y.data_ptr(), N, K // 2, N, bs, stream)
    assert rel_err(y.cpu(), O.gemm_4bit(x[:1], q_o, (N, K), am_o, bs, "nf4")[1]) < REL_TOL
    assert lib.get_context() is not None


def test_c_abi_reentrant_from_two_threads_on_two_streams():
    """the C entry points must be re-entrant from several Python threads (ctypes drops the GIL). Two threads, each
    on its own stream, hammer cgemm_4bit_bf16 - one in the streaming kernel's range (M = 1), one in the MFMA range with K
    slices (M = 24 on a narrow matrix: library-owned workspace, one per stream) - while a third flips the tuning knobs;
    every result must equal the single-threaded result bit for bit."""
    import threading

    from bitsandbytes_amd.cextension import lib

    F = _F()
    N, K, bs = 128, 4096, 64  # 8 column tiles: the MFMA launch splits K over two workgroups -> slabs in the library workspace
    W = (torch.randn(N, K, device=DEV) / K**0.5).bfloat16()
    q, st = F.quantize_4bit(W, blocksize=bs, quant_type="nf4")
    xs = {1: torch.randn(1, K, device=DEV).bfloat16(), 24: torch.randn(24, K, device=DEV).bfloat16()}
    assert lib.bnb_mi355x_gemm_4bit_workspace_bytes(0, 2, 24, N, K, bs) > 0
    ref = {}
    for M, x in xs.items():
        y = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
        lib.cgemm_4bit_bf16(x.data_ptr(), q.data_ptr(), st.absmax.data_ptr(), None, None, None, y.data_ptr(), None, M, N, K, bs,
                            2, torch._C._cuda_getCurrentRawStream(0))
        torch.cuda.synchronize()
        ref[M] = y.clone()
    errors = []
    stop = threading.Event()

    def worker(M):
        try:
            s = torch.cuda.Stream()
            x = xs[M]
            outs = [torch.empty(M, N, device=DEV, dtype=torch.bfloat16) for _ in range(8)]
            with torch.cuda.stream(s):
                for it in range(300):
                    y = outs[it % 8]
                    lib.cgemm_4bit_bf16(x.data_ptr(), q.data_ptr(), st.absmax.data_ptr(), None, None, None, y.data_ptr(), None,
                                        M, N, K, bs, 2, s.cuda_stream)
                    if it % 50 == 49:
                        s.synchronize()
                        for o in outs:
                            if not torch.equal(o, ref[M]):
                                errors.append(f"M={M} iteration {it}: result differs")
                s.synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    def knob_flipper():
        # production geometry <-> forced K slices of the same kernels: every setting is a correct geometry and, for a given
        # kernel family, the same summation order is NOT guaranteed across slice counts - so only the reserved knob is flipped
        while not stop.is_set():
            lib.bnb_mi355x_set_tuning(0, 0, 8, 0)  # (knob0 is reserved: same geometry, the stores race with the launches' snapshots)
            lib.bnb_mi355x_set_tuning(0, 0, 0, 0)

    threads = [threading.Thread(target=worker, args=(M,)) for M in (1, 24)]
    flip = threading.Thread(target=knob_flipper)
    flip.start()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    stop.set()
    flip.join()
    lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
    assert not errors, errors[:3]


@pytest.mark.parametrize("M", [1, 32])
def test_hip_graph_capture(M):
    """The fused op only enqueues kernels: it must be capturable and replayable."""
    import bitsandbytes_amd as bnb

    F = _F()
    N = K = 1024
    W = (torch.randn(N, K, device=DEV) / K**0.5).bfloat16()
    q, st = F.quantize_4bit(W, quant_type="nf4")
    x = torch.randn(M, K, device=DEV).bfloat16()
    y_eager = bnb.matmul_4bit(x, q, st)  # also warms up any lazily created workspace
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            bnb.matmul_4bit(x, q, st)
        with torch.cuda.graph(g, stream=s):
            y_graph = bnb.matmul_4bit(x, q, st)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y_graph, y_eager)


@pytest.mark.parametrize("double_quant", [False, True])
@pytest.mark.parametrize("storage", [torch.uint8, torch.bfloat16])
def test_linear4bit_module_gpu(double_quant, storage):
    from bitsandbytes_amd.nn import Linear4bit, Params4bit

    torch.manual_seed(3)
    ref = torch.nn.Linear(1024, 768, bias=True)
    layer = Linear4bit(1024, 768, bias=True, quant_type="nf4", compress_statistics=double_quant, quant_storage=storage,
                       compute_dtype=torch.bfloat16)
    layer.load_state_dict(ref.state_dict())
    layer = layer.to(DEV)  # quantises on the GPU
    assert layer.weight.bnb_quantized and layer.weight.dtype == storage and layer.weight.device.type == "cuda"
    for shape in ((1, 1024), (2, 7, 1024), (64, 1024)):
        x = torch.randn(*shape)
        y = layer(x.to(DEV))
        assert y.shape == (*shape[:-1], 768) and y.dtype == torch.float32
        y_o = _oracle_y(x.bfloat16().reshape(-1, 1024), layer.weight.data, layer.weight.quant_state,
                        layer.bias.detach().cpu().bfloat16())
        assert rel_err(y.reshape(-1, 768).cpu(), y_o) < REL_TOL
    # state-dict round trip -> identical outputs (reference tests/test_linear4bit.py:153-172)
    sd = {k: v.clone() for k, v in layer.state_dict().items()}
    new = Linear4bit(1024, 768, bias=True, quant_type="nf4", compress_statistics=double_quant, quant_storage=storage,
                     compute_dtype=torch.bfloat16)
    stats = {k[len("weight."):]: v for k, v in sd.items() if k.startswith("weight.")}
    new.weight = Params4bit.from_prequantized(sd["weight"], stats, device=DEV, module=new)
    new.bias.data = sd["bias"]
    new = new.to(DEV)
    x = torch.randn(4, 1024, device=DEV)
    assert torch.equal(new(x), layer(x))


def _oracle_grad_input(g, q, st):
    """grad_out @ dequantize_4bit(B) with the oracle's dequantize (rounded to the gradient dtype, like the reference's
    backward, autograd/_functions.py:384) and an fp64 product."""
    N, K = int(st.shape[0]), int(st.shape[1])
    if st.nested:
        absmax = O.dequantize_blockwise(st.absmax.cpu(), st.state2.absmax.cpu(), st.state2.code.cpu(), 256, torch.float32)
        absmax = absmax + st.offset.cpu()
    else:
        absmax = st.absmax.cpu()
    Wd = O.dequantize_4bit(q.cpu(), absmax.float(), st.blocksize, st.quant_type, (N, K), g.dtype)
    return (g.cpu().double() @ Wd.double()).float()


@pytest.mark.parametrize("M,N,K", [(1, 64, 128), (8, 256, 512), (64, 4096, 4096), (100, 1024, 11008), (200, 192, 384),
                                   (300, 128, 256)])
@pytest.mark.parametrize("variant", ["nf4-bs64", "nf4-bs64-nested", "fp4-bs128-nested-fp16", "nf4-bs256"])
def test_gemm_4bit_grad_input_fused_vs_oracle(M, N, K, variant):
    """The fused backward kernel (csrc/gemm4_grad_input.hip): grad_A = grad_out @ dequantize_4bit(B), every output against the
    oracle's dequantize + an fp64 product; several N slices, several 64-row passes, ragged M, batches above the fused range
    (the op's own dequantize + matmul fallback); bit-reproducible run to run."""
    F = _F()
    qt = "fp4" if variant.startswith("fp4") else "nf4"
    bs = 128 if "bs128" in variant else 256 if "bs256" in variant else 64
    dt = torch.float16 if "fp16" in variant else torch.bfloat16
    W = (torch.randn(N, K) / K**0.5).to(dt)
    g = torch.randn(M, N).to(dt)
    q, st = F.quantize_4bit(W.to(DEV), blocksize=bs, quant_type=qt, compress_statistics="nested" in variant)
    ref = _oracle_grad_input(g, q, st)
    if st.nested:
        args = (g.to(DEV), q, st.shape, st.state2.absmax, bs, qt, st.absmax, st.state2.code, st.offset)
    else:
        args = (g.to(DEV), q, st.shape, st.absmax, bs, qt)
    y1 = torch.ops.bitsandbytes_amd.gemm_4bit_grad_input.default(*args)
    y2 = torch.ops.bitsandbytes_amd.gemm_4bit_grad_input.default(*args)
    assert y1.shape == (M, K) and y1.dtype == dt
    assert rel_err(y1.cpu(), ref) < 4e-3   # one rounding of the result to 16 bits (the operands are the oracle's)
    assert torch.equal(y1, y2)


@pytest.mark.parametrize("M,N,K", [(16, 4096, 256), (17, 2240, 384), (32, 11008, 128), (33, 704, 512), (64, 64, 128), (128, 1088, 256)])
def test_gemm_4bit_grad_input_geometries(M, N, K):
    """The fused backward over 1 / 2 / 4 row tiles, N slices of unequal length, wavefronts with unequal block counts (N / 32
    not a multiple of the wavefront count), a slice shorter than a workgroup, two 64-row passes; plain and nested absmax."""
    F = _F()
    W = (torch.randn(N, K) / K**0.5).bfloat16()
    g = torch.randn(M, N).bfloat16()
    for nested in (False, True):
        q, st = F.quantize_4bit(W.to(DEV), blocksize=64, quant_type="nf4", compress_statistics=nested)
        ref = _oracle_grad_input(g, q, st)
        if nested:
            args = (g.to(DEV), q, st.shape, st.state2.absmax, 64, "nf4", st.absmax, st.state2.code, st.offset)
        else:
            args = (g.to(DEV), q, st.shape, st.absmax, 64, "nf4")
        y = torch.ops.bitsandbytes_amd.gemm_4bit_grad_input.default(*args)
        assert rel_err(y.cpu(), ref) < 4e-3, (nested, rel_err(y.cpu(), ref))
        assert torch.equal(y, torch.ops.bitsandbytes_amd.gemm_4bit_grad_input.default(*args))


def test_gemm_4bit_grad_input_exact_on_representable_inputs():
    """Gradients that are small integers and weights whose codes / scales are exactly representable: every product and sum
    is exact, so the fused kernel must equal the oracle bit for bit - an n paired with the wrong weight row, or a scale taken
    from a neighbouring block, cannot hide."""
    F = _F()
    M, N, K = 48, 256, 256
    g_ = torch.Generator().manual_seed(7)
    fp4 = F.get_4bit_type("fp4", device="cpu")
    allowed = torch.tensor([0, 3, 5, 7, 11, 13, 15])
    idx = allowed[torch.randint(0, len(allowed), (N, K), generator=g_)]
    idx[:, ::64] = 3
    scale = 2.0 ** torch.randint(-2, 3, (N, K // 64), generator=g_)
    W = (fp4[idx] * scale.repeat_interleave(64, dim=1)).to(torch.bfloat16)
    g = torch.randint(-4, 5, (M, N), generator=g_).to(torch.bfloat16)
    q, st = F.quantize_4bit(W.to(DEV), blocksize=64, quant_type="fp4")
    ref = _oracle_grad_input(g, q, st)
    y = torch.ops.bitsandbytes_amd.gemm_4bit_grad_input.default(g.to(DEV), q, st.shape, st.absmax, 64, "fp4")
    assert torch.equal(y.float().cpu(), ref.to(torch.bfloat16).float())


def test_gemm_4bit_grad_input_under_hip_graph_capture():
    """The fused backward is capture-safe (scratch from torch's allocator, no synchronisation, no library allocation) and
    replays to the same bits."""
    F = _F()
    M, N, K = 32, 512, 1024
    W = (torch.randn(N, K, device=DEV) / K**0.5).bfloat16()
    q, st = F.quantize_4bit(W, quant_type="nf4")
    g = torch.randn(M, N, device=DEV, dtype=torch.bfloat16)
    op = torch.ops.bitsandbytes_amd.gemm_4bit_grad_input.default
    eager = op(g, q, st.shape, st.absmax, 64, "nf4")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        op(g, q, st.shape, st.absmax, 64, "nf4")
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y = op(g, q, st.shape, st.absmax, 64, "nf4")
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, eager)


@pytest.mark.parametrize("M,N,K", [(8, 256, 512), (40, 200, 300)])
def test_backward_through_matmul_4bit_gpu(M, N, K):
    """MatMul4Bit.backward end to end (fused kernel for the first shape, the reference's dequantize + matmul formulation for
    the second, whose N and K are not whole tiles) against the oracle."""
    import bitsandbytes_amd as bnb

    F = _F()
    W = (torch.randn(N, K, device=DEV) / K**0.5).bfloat16()
    q, st = F.quantize_4bit(W, quant_type="nf4", blocksize=64)
    x = torch.randn(M, K, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # (K = 300 is not a multiple of the blocksize: the documented slow-path warning)
        y = bnb.matmul_4bit(x, q, st)
    g = torch.randn_like(y)
    y.backward(g)
    assert x.grad.shape == x.shape and x.grad.dtype == x.dtype
    assert rel_err(x.grad.detach().cpu(), _oracle_grad_input(g, q, st)) < 4e-3


def test_backward_above_the_fused_range_with_nested_statistics():
    """QLoRA-shaped backward (double quantisation, M in the hundreds): dequantize - statistics reconstructed inside that ONE launch -
    + library matmul. Same bits as the same operator handed the reconstructed fp32 absmax; and end to end against the oracle."""
    import bitsandbytes_amd as bnb

    F = _F()
    torch.manual_seed(11)
    N, K, M = 768, 1024, 384
    W = (torch.randn(N, K, device=DEV) / K**0.5).bfloat16()
    q, st = F.quantize_4bit(W, quant_type="nf4", blocksize=64, compress_statistics=True)
    g = torch.randn(M, N, device=DEV, dtype=torch.bfloat16)
    op = torch.ops.bitsandbytes_amd.gemm_4bit_grad_input.default
    am = torch.ops.bitsandbytes.dequantize_blockwise.default(st.absmax, st.state2.absmax, st.state2.code, 256, torch.float32) + st.offset
    a = op(g, q, st.shape, st.state2.absmax, 64, "nf4", absmax_8bit=st.absmax, absmax_code=st.state2.code, absmax_offset=st.offset)
    b = op(g, q, st.shape, am, 64, "nf4")
    assert torch.equal(a, b)
    x = torch.randn(M, K, device=DEV, dtype=torch.bfloat16, requires_grad=True)
    bnb.matmul_4bit(x, q, st).backward(g)
    assert rel_err(x.grad.detach().cpu(), _oracle_grad_input(g, q, st)) < 4e-3


# kernel families as bnb_mi355x_last_gemm_kernel() reports them (include/bnb_mi355x.h)
K_STREAM, K_GENERIC, K_RT, K_PC, K_KQ, K_SM = 1, 2, 3, 4, 6, 7


class _forced:
    """`with _forced(knob, K_xx): ...` - run with a forced MFMA geometry (bnb_mi355x_set_tuning knob1) and assert on exit that the
    kernel family the test names is the one the last call LAUNCHED: a forced kernel that does not serve a case falls back to another
    family by design, and a test that passes on the fall-back is not coverage of the kernel in its name."""

    def __init__(self, knob, expect):
        self.knob, self.expect = knob, expect

    def __enter__(self):
        import bitsandbytes_amd as bnb

        bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, self.knob)
        return self

    def __exit__(self, et, ev, tb):
        import bitsandbytes_amd as bnb

        bnb.lib.bnb_mi355x_set_tuning(0, 0, 0, 0)
        if et is None and self.expect is not None:
            ran = bnb.lib.bnb_mi355x_last_gemm_kernel()
            assert ran == self.expect, f"knob {self.knob}: kernel family {ran} ran, the test is about {self.expect}"
        return False


@pytest.mark.parametrize("cfg", [11, 12, 13, 14])
@pytest.mark.parametrize("M,N,K,ks", [(5, 256, 1024, 1), (16, 200, 2048, 2), (33, 384, 1024, 1), (64, 512, 4096, 4),
                                      (64, 1000, 2816, 1), (100, 128, 512, 2)])
def test_mfma_kernel_variants(cfg, M, N, K, ks):
    """Every geometry of the producer/consumer MFMA kernel (8x1, 4x2, 8x2, 4x1 consumers x n-tiles), incl. cross-workgroup K
    slices and ragged N / M, against the oracle; results must also be bit-reproducible run to run."""
    import bitsandbytes_amd as bnb

    F = _F()
    W = (torch.randn(N, K) / K**0.5).bfloat16()
    x = torch.randn(M, K).bfloat16()
    bias = torch.randn(N).bfloat16()
    for dq in (False, True):
        q, st = F.quantize_4bit(W.to(DEV), blocksize=64, quant_type="nf4", compress_statistics=dq)
        y_ref = _oracle_y(x, q, st, bias)
        with _forced(cfg * 100 + ks, K_PC):
            y1 = _run_kernel(2, x.to(DEV), q, st, bias.to(DEV))
            y2 = _run_kernel(2, x.to(DEV), q, st, bias.to(DEV))
        assert rel_err(y1.cpu(), y_ref) < REL_TOL
        assert torch.equal(y1, y2)


@pytest.mark.parametrize("cfg,ks", [(20, 0), (21, 0), (22, 0), (20, 2), (22, 3)])
@pytest.mark.parametrize("M,N,K", [(3, 256, 1024), (5, 200, 2048), (8, 4096, 4096), (13, 96, 4352), (16, 1376, 4096),
                                   (16, 512, 11008 - 11008 % 256), (33, 384, 1024), (64, 512, 4096), (100, 130, 512)])
def test_mfma_rt_kernel_geometries(cfg, ks, M, N, K):
    """The register-transposed MFMA kernel (csrc/gemm4_mfma_rt.hip) in every launch geometry - built-in / 8 / 16
    wavefronts, forced K slices, ragged N and M, wavefronts with no / one / several chunks - against the oracle,
    for fp32 and nested absmax, NF4 and FP4, blocksize 64 and 128, bf16 and fp16; bit-reproducible run to run."""
    import bitsandbytes_amd as bnb

    F = _F()
    for dtype, qt, bs, dq in ((torch.bfloat16, "nf4", 64, False), (torch.bfloat16, "nf4", 64, True),
                              (torch.float16, "fp4", 128, True), (torch.float16, "nf4", 256, False)):
        W = (torch.randn(N, K) / K**0.5).to(dtype)
        x = torch.randn(M, K).to(dtype)
        bias = torch.randn(N).to(dtype)
        q, st = F.quantize_4bit(W.to(DEV), blocksize=bs, quant_type=qt, compress_statistics=dq)
        y_ref = _oracle_y(x, q, st, bias)
        with _forced(cfg * 100 + ks, K_RT):
            y1 = _run_kernel(2, x.to(DEV), q, st, bias.to(DEV))
            y2 = _run_kernel(2, x.to(DEV), q, st, bias.to(DEV))
            y3 = _run_kernel(2, x.to(DEV), q, st, None)
        assert rel_err(y1.cpu(), y_ref) < REL_TOL, (dtype, qt, bs, dq)
        assert torch.equal(y1, y2)
        assert rel_err(y3.cpu(), _oracle_y(x, q, st, None)) < REL_TOL


@pytest.mark.parametrize("cfg,ks", [(40, 0), (40, 1), (40, 2), (40, 3)])
@pytest.mark.parametrize("M,N,K", [(5, 256, 1024), (17, 200, 2048), (32, 4096, 4096), (33, 384, 1024), (64, 512, 4096),
                                   (64, 1000, 2816), (100, 130, 512), (128, 1376, 4096), (200, 256, 256), (17, 512, 11008 - 11008 % 256)])
def test_mfma_kq_kernel_geometries(cfg, ks, M, N, K):
    """The K-quarter MFMA kernel (csrc/gemm4_mfma_kq.hip) - one and two 32-row tiles, several row passes, built-in and forced K
    slices (incl. slices of unequal length, one-chunk slices and the accumulator-layout slabs of gemm4_finalize_kq_kernel), ragged N
    and M - against the oracle, for fp32 and nested absmax, NF4 and FP4, blocksize 64 / 128 / 256, bf16 and fp16; bit-reproducible
    run to run. Nested statistics at a blocksize other than 64 are not served by this kernel (gemm_4bit_kq_serves): that case
    must run - and is asserted to run - the producer/consumer kernel."""
    F = _F()
    for dtype, qt, bs, dq in ((torch.bfloat16, "nf4", 64, False), (torch.bfloat16, "nf4", 64, True),
                              (torch.float16, "fp4", 128, True), (torch.float16, "nf4", 256, False), (torch.float16, "fp4", 64, True)):
        W = (torch.randn(N, K) / K**0.5).to(dtype)
        x = torch.randn(M, K).to(dtype)
        bias = torch.randn(N).to(dtype)
        q, st = F.quantize_4bit(W.to(DEV), blocksize=bs, quant_type=qt, compress_statistics=dq)
        y_ref = _oracle_y(x, q, st, bias)
        with _forced(cfg * 100 + ks, K_PC if (dq and bs != 64) else K_KQ):
            y1 = _run_kernel(2, x.to(DEV), q, st, bias.to(DEV))
            y2 = _run_kernel(2, x.to(DEV), q, st, bias.to(DEV))
            y3 = _run_kernel(2, x.to(DEV), q, st, None)
        assert rel_err(y1.cpu(), y_ref) < REL_TOL, (dtype, qt, bs, dq)
        assert torch.equal(y1, y2)
        assert rel_err(y3.cpu(), _oracle_y(x, q, st, None)) < REL_TOL


@pytest.mark.parametrize("M,N,K", [(1, 14336, 4096), (2, 4096, 4096), (4, 1376, 4096), (8, 4096, 4096), (16, 512, 2816), (48, 4096, 4096),
                                   (64, 8192, 8192), (200, 1024, 2048)])
@pytest.mark.parametrize("dtype,qt,bs", [(torch.bfloat16, "nf4", 64), (torch.float16, "fp4", 128), (torch.bfloat16, "nf4", 32)])
def test_nested_statistics_equal_the_reconstructed_fp32_absmax_bit_for_bit_in_every_kernel(M, N, K, dtype, qt, bs):
    """The scale of a double-quantised block is code2[q] * absmax2 + offset with TWO roundings in every kernel - the value the host-side
    sequence (dequantize_blockwise, += offset) hands to the same kernel as plain statistics. So a call with nested statistics and the
    call with the reconstructed fp32 absmax give the same bits: forward in the routed kernel and in each forced family, and the fused
    backward. (Until round 5 hipcc had contracted the kernels' expression into one fma - a last-bit difference in ~1/5 of the scales,
    invisible to a tolerance, visible as a flipped bf16 output every few calls of a 14336-row layer; the sharded layers carry
    un-nested statistics and promise the unsharded layer's bits.)"""
    from bitsandbytes_amd.backends import hip

    F = _F()
    torch.manual_seed(M * 7 + bs)
    W = (torch.randn(N, K, device=DEV) / K**0.5).to(dtype)
    q, st = F.quantize_4bit(W, blocksize=bs, quant_type=qt, compress_statistics=True)
    am = torch.ops.bitsandbytes.dequantize_blockwise.default(st.absmax, st.state2.absmax, st.state2.code, 256, torch.float32) + st.offset
    nested = (st.absmax, st.state2.code, st.offset)
    import bitsandbytes_amd as bnb

    compared = 0
    for kernel in (0, 2, 3):
        for it in range(3):
            x = (torch.randn(M, K, device=DEV) * (1 + it)).to(dtype)
            a = hip._gemm_4bit_fused(x, q, st.shape, st.state2.absmax, bs, qt, None, *nested, kernel=kernel)
            fam_a = bnb.lib.bnb_mi355x_last_gemm_kernel()
            b = hip._gemm_4bit_fused(x, q, st.shape, am, bs, qt, None, None, None, None, kernel=kernel)
            fam_b = bnb.lib.bnb_mi355x_last_gemm_kernel()
            if fam_a != fam_b:
                # (the router may serve the two kinds of statistics with different families - the K-quarter kernel takes nested ones at
                # blocksize 64 only, the MFMA route at blocksize 32 plain ones only: another summation order, nothing to compare)
                assert rel_err(a, b.float()) < 1e-2
                continue
            compared += 1
            assert torch.equal(a, b), (kernel, it, fam_a)
    assert compared >= 3  # (kernel = 3, the streaming kernel, serves both kinds everywhere)
    if M <= 128 and bs >= 64:
        g = torch.randn(M, N, device=DEV).to(dtype)
        op = torch.ops.bitsandbytes_amd.gemm_4bit_grad_input.default
        a = op(g, q, st.shape, st.state2.absmax, bs, qt, absmax_8bit=st.absmax, absmax_code=st.state2.code, absmax_offset=st.offset)
        b = op(g, q, st.shape, am, bs, qt)
        assert torch.equal(a, b)


@pytest.mark.parametrize("mis", [1, 2, 3])
def test_mfma_kq_kernel_nested_codes_need_dword_alignment_or_fall_back(mis):
    """The K-quarter kernel fetches a column's four nested 8-bit codes of a chunk as ONE aligned dword; a row shard's view of the
    code array can start at any byte - such a call must fall back to the producer/consumer kernel and still be right."""
    from bitsandbytes_amd.backends import hip

    F = _F()
    M, N, K = 40, 260, 1024
    W = (torch.randn(N, K) / K**0.5).bfloat16()
    x = torch.randn(M, K).bfloat16().to(DEV)
    q, st = F.quantize_4bit(W.to(DEV), blocksize=64, quant_type="nf4", compress_statistics=True)
    buf = torch.zeros(st.absmax.numel() + 8, dtype=torch.uint8, device=DEV)
    a8 = buf[mis:mis + st.absmax.numel()]
    a8.copy_(st.absmax)
    assert a8.data_ptr() % 4 == mis
    with _forced(4000, K_PC):
        y = hip._gemm_4bit_fused(x, q, st.shape, st.state2.absmax, st.blocksize, st.quant_type, None, a8, st.state2.code,
                                 st.offset, kernel=2)
    with _forced(4000, K_KQ):
        y0 = _run_kernel(2, x, q, st, None)
    assert rel_err(y.cpu(), _oracle_y(x.cpu(), q, st, None)) < REL_TOL
    assert rel_err(y0.cpu(), _oracle_y(x.cpu(), q, st, None)) < REL_TOL


@pytest.mark.parametrize("N,K,M,dq", [(8192, 8192, 64, False), (8192, 8192, 64, True), (11008, 4096, 48, True), (4096, 11008 - 11008 % 256, 17, False),
                                      (6144, 4096, 64, True), (4096, 4096, 300, True), (11008, 4096, 512, False)])
def test_mfma_kq_kernel_is_deterministic_on_the_shapes_the_router_gives_it(N, K, M, dq):
    """Round-3 review: a kernel family with LDS-DMA rings and counted waits is only routed by default with a repeated-launch
    determinism + parity check on the shapes the router actually picks (not only on small forced ones): 40 launches of the
    BUILT-IN route, asserted to be the K-quarter kernel, all bit-identical, the first within tolerance of a fp32 dequantize +
    matmul on the device (the oracle comparison at these sizes is test_config3_8192_all_rows / the BASELINE config tests)."""
    import bitsandbytes_amd as bnb

    F = _F()
    g = torch.Generator(device=DEV).manual_seed(N + K + M)
    W = (torch.randn(N, K, device=DEV, generator=g) / K**0.5).bfloat16()
    q, st = F.quantize_4bit(W, blocksize=64, quant_type="nf4", compress_statistics=dq)
    x = torch.randn(M, K, device=DEV, generator=g).bfloat16()
    ref = x.float() @ F.dequantize_4bit(q, st).float().t()
    y0 = _run_kernel(0, x, q, st, None).clone()
    assert bnb.lib.bnb_mi355x_last_gemm_kernel() == K_KQ
    assert rel_err(y0.float().cpu(), ref.cpu()) < REL_TOL
    # (other launches in between move the workgroups' timing around: a race that needs a particular interleaving gets its chance)
    junk = torch.randn(1 << 20, device=DEV)
    for i in range(40):
        if i % 3 == 0:
            junk = junk * 1.0001
        y = _run_kernel(0, x, q, st, None)
        assert torch.equal(y, y0), f"launch {i} differs from the first"


def test_mfma_kernels_exact_on_representable_inputs():
    """Activations that are small integers and weights whose codes / scales are exactly representable make every product
    and every partial sum exact in fp32: the kernel must then equal the oracle bit for bit - a k that is paired with the
    wrong weight, or a block that gets its neighbour's scale, cannot hide inside the 1e-2 tolerance."""
    F = _F()
    M, N, K = 16, 64, 1024
    g = torch.Generator().manual_seed(5)
    # FP4 codes that bf16 holds exactly: 0, 1, 0.5, 0.25 and their negatives (indices 0, 3, 5, 7, 8, 11, 13, 15); with a
    # power-of-two absmax per block (first element of every block = code 1.0) quantization reproduces the indices exactly
    fp4 = F.get_4bit_type("fp4", device="cpu")
    allowed = torch.tensor([0, 3, 5, 7, 11, 13, 15])
    idx = allowed[torch.randint(0, len(allowed), (N, K), generator=g)]
    idx[:, ::64] = 3
    scale = 2.0 ** torch.randint(-2, 3, (N, K // 64), generator=g)
    W = (fp4[idx] * scale.repeat_interleave(64, dim=1)).to(torch.bfloat16)
    x = torch.randint(-4, 5, (M, K), generator=g).to(torch.bfloat16)
    q, st = F.quantize_4bit(W.to(DEV), blocksize=64, quant_type="fp4")
    y_ref = _oracle_y(x, q, st, None)
    import bitsandbytes_amd as bnb
    # register-transposed kernel (one / two K slices), producer/consumer kernel, K-quarter kernel (one / two K slices)
    for knob, fam in ((2000, K_RT), (2002, K_RT), (1100, K_PC), (4000, K_KQ), (4002, K_KQ), (5000, K_SM)):
        with _forced(knob, fam):
            y = _run_kernel(2, x.to(DEV), q, st, None)
        assert torch.equal(y.float().cpu(), y_ref.to(torch.bfloat16).float()), knob


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "fp16"])
@pytest.mark.parametrize("M", [3, 5, 8, 16, 17, 33, 64, 100, 200])
def test_gemm_4bit_blocksize_32_runs_the_mfma_kernel(M, dtype):
    """Round 5: blocksize 32 on the MFMA route (the register-transposed kernel's BS32 instances - a full 4 x 4 transposition of the
    weight dwords between lane groups so that every MFMA consumes ONE 32-k block; reference capability: any power-of-two blocksize,
    csrc/gemm_4bit_simt.cu:208,225). Every row-tile geometry (direct fragments, 1 - 4 row tiles, row passes over grid.z), ragged
    N, NF4 / FP4, with bias - against the oracle, and the family that RAN is asserted. Nested statistics at blocksize 32 keep the
    streaming kernel (same values, checked too)."""
    import bitsandbytes_amd as bnb

    F = _F()
    for (N, K, qt) in ((512, 1024, "nf4"), (1000, 2816, "fp4"), (4096, 4096, "nf4")):
        if N * K > (4 << 20) and M not in (5, 16, 64):
            continue
        g = torch.Generator().manual_seed(M + N)
        W = (torch.randn(N, K, generator=g) / K**0.5).to(dtype)
        x = torch.randn(M, K, generator=g).to(dtype)
        bias = torch.randn(N, generator=g).to(dtype)
        q, st = F.quantize_4bit(W.to(DEV), blocksize=32, quant_type=qt)
        from bitsandbytes_amd.backends import hip

        if hip._gemm_4bit_route(dtype, M, N, K, 32, False) == "fused":
            call = lambda: bnb.matmul_4bit(x.to(DEV), q, st, bias=bias.to(DEV))  # noqa: E731
        else:
            # above FUSED_MAX_M_BS32 rows the public route is dequantize + GEMM (cheaper from ~200 rows on: the BS32 instances' row
            # passes run one after the other); the kernel's row-pass geometry is still exercised here, through the library call
            assert M > hip.FUSED_MAX_M_BS32
            y_pub = bnb.matmul_4bit(x.to(DEV), q, st, bias=bias.to(DEV))
            assert rel_err(y_pub.cpu(), _oracle_y(x, q, st, bias)) < REL_TOL
            call = lambda: _run_kernel(0, x.to(DEV), q, st, bias.to(DEV))  # noqa: E731
        y = call()
        # (three and four rows go to the MFMA kernels on matrices of >= 12 M weights only: c_api.hip route_to_mfma)
        want = K_RT if M > 4 or N * K >= (12 << 20) else K_STREAM
        assert bnb.lib.bnb_mi355x_last_gemm_kernel() == want, (M, N, K, bnb.lib.bnb_mi355x_last_gemm_kernel())
        y_ref = _oracle_y(x, q, st, bias)
        assert rel_err(y.cpu(), y_ref) < REL_TOL, (M, N, K, qt)
        assert torch.equal(y, call())  # bit-reproducible
    # double quantisation at blocksize 32: outside the BS32 instances - the streaming kernel (4-row passes) up to STREAM_ONLY_MAX_M rows,
    # dequantize + library GEMM above (3 - 4 x cheaper than the passes at 64 rows: profiles/r5_tall_small_ab.txt); same tolerance
    from bitsandbytes_amd.backends import hip

    q, st = F.quantize_4bit(W.to(DEV), blocksize=32, quant_type="nf4", compress_statistics=True)
    y = bnb.matmul_4bit(x.to(DEV), q, st)
    if M <= hip.STREAM_ONLY_MAX_M:
        assert bnb.lib.bnb_mi355x_last_gemm_kernel() == K_STREAM
    else:
        assert hip._gemm_4bit_route(dtype, M, st.shape[0], st.shape[1], 32, True) == "unfused"
        assert bnb.lib.bnb_mi355x_last_gemm_kernel() == K_RT  # (the plain-statistics call above: this one launched no fused kernel)
    assert rel_err(y.cpu(), _oracle_y(x, q, st)) < REL_TOL


def test_mfma_blocksize_32_exact_on_representable_inputs():
    """The blocksize-32 twin of the test above: with exactly representable codes, power-of-two scales per 32-k block and small
    integer activations every product and partial sum is exact - a k multiplied with the wrong weight, or a 32-k block scaled with
    its neighbour's absmax (what a wrong lane-group transposition would do), cannot hide inside 1e-2."""
    F = _F()
    import bitsandbytes_amd as bnb

    for M in (5, 16, 48):  # (from five rows on every matrix takes the MFMA route: c_api.hip route_to_mfma)
        N, K = 80, 1024
        g = torch.Generator().manual_seed(7 + M)
        fp4 = F.get_4bit_type("fp4", device="cpu")
        allowed = torch.tensor([0, 3, 5, 7, 11, 13, 15])
        idx = allowed[torch.randint(0, len(allowed), (N, K), generator=g)]
        idx[:, ::32] = 3  # code 1.0 at the head of every 32-k block: its absmax is the block's scale
        scale = 2.0 ** torch.randint(-3, 4, (N, K // 32), generator=g)
        W = (fp4[idx] * scale.repeat_interleave(32, dim=1)).to(torch.bfloat16)
        x = torch.randint(-4, 5, (M, K), generator=g).to(torch.bfloat16)
        q, st = F.quantize_4bit(W.to(DEV), blocksize=32, quant_type="fp4")
        assert torch.equal(st.absmax.cpu().view(N, K // 32), scale.float())
        y = bnb.matmul_4bit(x.to(DEV), q, st)
        assert bnb.lib.bnb_mi355x_last_gemm_kernel() == K_RT
        y_ref = _oracle_y(x, q, st, None)
        assert torch.equal(y.float().cpu(), y_ref.to(torch.bfloat16).float()), M


def test_blocksize_32_call_with_a_caller_supplied_code_table_runs_the_streaming_kernel():
    """the blocksize-32 MFMA route accepted calls its BS32 instances then refused - a public C-ABI call
    with a caller-supplied code table (`code16`) at blocksize 32 printed 'internal error' and ended the process. The route now asks the
    instances' own preconditions (gemm_4bit_rt_supported) and such a call runs the streaming kernel: any M, kernel = 0 and the forced
    MFMA request alike, values against the oracle."""
    import bitsandbytes_amd as bnb
    from bitsandbytes_amd.backends import hip

    F = _F()
    N, K = 512, 1024
    g = torch.Generator().manual_seed(3)
    W = (torch.randn(N, K, generator=g) / K**0.5).bfloat16()
    q, st = F.quantize_4bit(W.to(DEV), blocksize=32, quant_type="nf4")
    code16 = F.get_4bit_type("nf4", device=DEV)
    for M in (1, 8, 40):
        x = torch.randn(M, K, generator=g).bfloat16()
        for kernel in (0, 2):
            y = hip._gemm_4bit_fused(x.to(DEV), q, st.shape, st.absmax, 32, "nf4", None, None, None, None, kernel=kernel, code16=code16)
            assert bnb.lib.bnb_mi355x_last_gemm_kernel() == K_STREAM, (M, kernel)
            assert rel_err(y.cpu(), _oracle_y(x, q, st, None)) < REL_TOL, (M, kernel)


# ------------------------------------------------------------------------------------------ streaming MFMA kernel (round 6)
@pytest.mark.parametrize("M", [1, 2, 3, 4, 5, 7, 8, 9, 13, 16, 17, 33])
@pytest.mark.parametrize("N,K", [(16, 256), (200, 512), (4100, 512), (5000, 1024), (12345, 768), (20000, 256), (70000, 512), (4097, 8192),
                                 (600, 11008 - 11008 % 256), (4096, 2752), (3100, 1344), (200, 1088), (4100, 320), (5000, 64), (48, 192), (300, 4096 + 128)])
def test_mfma_sm_kernel_geometries(M, N, K):
    """The streaming MFMA kernel (csrc/gemm4_mfma_sm.hip), forced, in every launch geometry: 4 / 8 / 16 staged activation rows
    (16 / 8 wavefronts), row passes above 16 rows, one to four tiles per workgroup and several rounds of workgroups (N = 70000),
    single-item and ring instances, wavefronts with no / one / several chunks (K = 256 ... 10752), rows that are not whole 256-k
    chunks (K = 2752, 1344, 1088, 320, 192, 64: the last chunk holds one to three 64-k blocks - the reference's fused kernels take any
    K % blocksize == 0, csrc/gemm_4bit_simt.cu:208,225), ragged N and tile rows past the workgroup's share - against the oracle, for fp32 and nested absmax, NF4 and FP4, blocksize 64 ... 512, bf16 and fp16, with and
    without bias; bit-reproducible run to run; the family that ran is asserted."""
    F = _F()
    if N * K > (8 << 20) and M not in (2, 8, 16):
        pytest.skip("large shapes: three batch sizes")
    for dtype, qt, bs, dq in ((torch.bfloat16, "nf4", 64, False), (torch.bfloat16, "nf4", 64, True), (torch.float16, "fp4", 128, True),
                              (torch.float16, "nf4", 256, False), (torch.bfloat16, "fp4", 512, True)):
        if (N * K) % bs or K % bs:
            continue
        g = torch.Generator().manual_seed(M + N + K)
        W = (torch.randn(N, K, generator=g) / K**0.5).to(dtype)
        x = torch.randn(M, K, generator=g).to(dtype)
        bias = torch.randn(N, generator=g).to(dtype)
        q, st = F.quantize_4bit(W.to(DEV), blocksize=bs, quant_type=qt, compress_statistics=dq)
        y_ref = _oracle_y(x, q, st, bias)
        with _forced(5000, K_SM):
            y1 = _run_kernel(2, x.to(DEV), q, st, bias.to(DEV))
            y2 = _run_kernel(2, x.to(DEV), q, st, bias.to(DEV))
            y3 = _run_kernel(2, x.to(DEV), q, st, None)
        assert rel_err(y1.cpu(), y_ref) < REL_TOL, (dtype, qt, bs, dq)
        assert torch.equal(y1, y2)
        assert rel_err(y3.cpu(), _oracle_y(x, q, st, None)) < REL_TOL
        # every row on its own: a row of the batch mixed with another row's activations cannot hide in the batch's norm
        worst = ((y1.double().cpu() - y_ref.double()).norm(dim=1) / y_ref.double().norm(dim=1)).max()
        assert worst < 2 * REL_TOL, (dtype, qt, bs, dq, float(worst))


def test_mfma_sm_kernel_large_blocksizes_and_blocks_that_span_rows():
    """Quantization blocks are blocks of the FLAT weight tensor (reference functional.py:925-936): with K % blocksize != 0 a block
    ends in one row and continues in the next, with blocksize > K one block covers several rows. The Python route sends such calls to
    dequantize + linear as the reference does (backends/cuda/ops.py:956-962), the C ABI takes them: the streaming MFMA kernel indexes
    the statistics by flat 64-k group, so every blocksize from 64 to 4096 and every K % 64 == 0 is the same code path. Forced and
    routed, fp32 and nested statistics, against the oracle."""
    import bitsandbytes_amd as bnb

    F = _F()
    for (N, K, bs) in ((3200, 192, 128), (3200, 320, 256), (3072, 64, 4096), (4096, 1088, 2048), (3328, 8192, 4096), (3200, 4096, 1024), (200, 704, 512)):
        assert (N * K) % bs == 0
        for dq in (False, True):
            g = torch.Generator().manual_seed(N + K + bs)
            W = (torch.randn(N, K, generator=g) / K**0.5).to(torch.bfloat16)
            q, st = F.quantize_4bit(W.to(DEV), blocksize=bs, quant_type="nf4", compress_statistics=dq)
            for M in (2, 7, 16):
                x = torch.randn(M, K, generator=g).to(torch.bfloat16)
                y_ref = _oracle_y_full(x, q, st)
                with _forced(5000, K_SM):
                    y = _run_kernel(2, x.to(DEV), q, st)
                assert rel_err(y.cpu(), y_ref) < REL_TOL, (N, K, bs, dq, M)
                y0 = _run_kernel(0, x.to(DEV), q, st)  # the library's own route
                if N >= 3072 and M <= 8:
                    assert bnb.lib.bnb_mi355x_last_gemm_kernel() == K_SM, (N, K, bs, M)
                assert rel_err(y0.cpu(), y_ref) < REL_TOL, (N, K, bs, dq, M)


def test_mfma_sm_kernel_exact_on_representable_inputs():
    """Small-integer activations, exactly representable FP4 codes and a different power-of-two scale per 64-k block: every product
    and partial sum is exact in fp32, so the streaming MFMA kernel equals the oracle bit for bit - for every staged-row count, for
    single-item and ring instances, several tiles per workgroup and several chunks per wavefront. A fragment piece read from the wrong
    slot of the swizzled staging area, a k paired with the wrong weight or a block scaled with its neighbour's absmax cannot hide."""
    F = _F()
    fp4 = F.get_4bit_type("fp4", device="cpu")
    allowed = torch.tensor([0, 3, 5, 7, 11, 13, 15])
    for (N, K) in ((80, 1024), (4096 + 48, 4096), (8192, 8192), (2048, 11008 - 11008 % 256), (4096, 2752), (100, 4096 + 192), (64, 64)):
        g = torch.Generator().manual_seed(N + K)
        idx = allowed[torch.randint(0, len(allowed), (N, K), generator=g)]
        idx[:, ::64] = 3  # code 1.0 at the head of every block: its absmax is the block's scale
        scale = 2.0 ** torch.randint(-2, 3, (N, K // 64), generator=g)
        W = (fp4[idx] * scale.repeat_interleave(64, dim=1)).to(torch.bfloat16)
        q, st = F.quantize_4bit(W.to(DEV), blocksize=64, quant_type="fp4")
        assert torch.equal(st.absmax.cpu().view(N, K // 64), scale.float())
        Wd = F.dequantize_4bit(q, st).double().cpu()
        for M in (2, 3, 4, 7, 8, 11, 16, 20):
            # (distinct rows: row m carries a pattern no other row has)
            x = torch.randint(-4, 5, (M, K), generator=g).to(torch.bfloat16)
            with _forced(5000, K_SM):
                y = _run_kernel(2, x.to(DEV), q, st, None)
            y_ref = (x.double() @ Wd.t()).to(torch.bfloat16)
            assert torch.equal(y.float().cpu(), y_ref.float()), (N, K, M)


@pytest.mark.parametrize("N,K,dq", [(4096, 4096, False), (4096, 4096, True), (8192, 8192, False), (11008, 4096, True), (14336, 4096, False),
                                    (4096, 11008 - 11008 % 256, True), (5120, 5120, False), (4096, 2752, False), (4096, 2752, True)])
def test_mfma_sm_kernel_is_routed_and_deterministic(N, K, dq):
    """The BUILT-IN route takes the streaming MFMA kernel for 2 ... 16 rows on these matrices (one row: the streaming
    kernel; long rows with more than 8 batch rows: the register-transposed kernel). 30 launches each with other work in between, all
    bit-identical; the first within tolerance of fp32 dequantize + matmul on the device."""
    import bitsandbytes_amd as bnb

    F = _F()
    g = torch.Generator(device=DEV).manual_seed(N + K)
    W = (torch.randn(N, K, device=DEV, generator=g) / K**0.5).bfloat16()
    q, st = F.quantize_4bit(W, blocksize=64, quant_type="nf4", compress_statistics=dq)
    Wd = F.dequantize_4bit(q, st).float()
    junk = torch.randn(1 << 20, device=DEV)
    for M in (1, 2, 4, 8, 9, 16, 17, 40, 64):
        x = torch.randn(M, K, device=DEV, generator=g).bfloat16()
        y0 = bnb.matmul_4bit(x, q, st).clone()
        fam = bnb.lib.bnb_mi355x_last_gemm_kernel()
        if K % 256:  # (rows that are not whole 256-k chunks: the streaming MFMA kernel's row passes up to 64 rows)
            want = K_STREAM if M == 1 else K_SM
        else:
            want = K_STREAM if M == 1 else K_SM if M <= 8 or (M <= 16 and K <= 2 * N) else None
        from bitsandbytes_amd.backends import hip

        if want is not None:
            assert fam == want, (M, fam)
        elif hip._gemm_4bit_route(torch.bfloat16, M, N, K, 64, dq) == "fused":  # (else: dequantize + GEMM, no fused launch to ask about)
            assert fam != K_SM
        assert rel_err(y0.float().cpu(), (x.float() @ Wd.t()).cpu()) < REL_TOL
        for i in range(30):
            if i % 3 == 0:
                junk = junk * 1.0001
            assert torch.equal(bnb.matmul_4bit(x, q, st), y0), (M, i)



def test_mfma_tall_tile_experiment_is_correct_and_deterministic():
    """csrc/gemm4_mfma_tall.hip (round 6, review item 2; reachable with knob cfg 60, NOT routed: 27.6 % of the dense peak against a bar of 40,
    profiles/r6_tall_tile_experiment.txt): 128 x 128 tiles, the pre-scaled operand T(code * scale) decoded once per 128 rows, LDS-DMA
    activation ring, hand-counted waits. Kept in the library as the measured starting point of that work - and therefore tested: against
    the oracle over ragged M / N, several step counts (incl. fewer steps than the ring is deep), nested statistics, FP4, blocksize 64 ... 256,
    both dtypes, with bias; bit-exact on exactly-representable inputs; bit-reproducible run to run. (K in whole 256-k chunks: the forced MFMA
    request - kernel = 2 - is only honoured where the MFMA family's own precondition holds.)"""
    F = _F()
    K_TALL = 8
    for (M, N, K, dtype, qt, bs, dq) in ((128, 256, 512, torch.bfloat16, "nf4", 64, False), (300, 200, 1024, torch.float16, "fp4", 128, True),
                                         (130, 1000, 256, torch.bfloat16, "nf4", 64, True), (1024, 4096, 4096, torch.bfloat16, "nf4", 64, False),
                                         (257, 384, 768, torch.bfloat16, "fp4", 64, False), (512, 512, 2816, torch.float16, "nf4", 256, False)):
        g = torch.Generator().manual_seed(M + N + K)
        W = (torch.randn(N, K, generator=g) / K**0.5).to(dtype)
        x = torch.randn(M, K, generator=g).to(dtype)
        bias = torch.randn(N, generator=g).to(dtype)
        q, st = F.quantize_4bit(W.to(DEV), blocksize=bs, quant_type=qt, compress_statistics=dq)
        with _forced(6000, K_TALL):
            y1 = _run_kernel(2, x.to(DEV), q, st, bias.to(DEV))
            y2 = _run_kernel(2, x.to(DEV), q, st, bias.to(DEV))
        ref = (x.double().to(DEV) @ F.dequantize_4bit(q, st).double().t() + bias.double().to(DEV))
        assert float((y1.double() - ref).norm() / ref.norm()) < REL_TOL, (M, N, K)
        worst = ((y1.double() - ref).norm(dim=1) / ref.norm(dim=1)).max()
        assert worst < 2 * REL_TOL, (M, N, K, float(worst))
        assert torch.equal(y1, y2)
    # exact inputs: small-integer activations, exactly representable FP4 codes, a power-of-two scale per block
    fp4 = F.get_4bit_type("fp4", device="cpu")
    allowed = torch.tensor([0, 3, 5, 7, 11, 13, 15])
    M, N, K = 200, 300, 1024
    g = torch.Generator().manual_seed(9)
    idx = allowed[torch.randint(0, len(allowed), (N, K), generator=g)]
    idx[:, ::64] = 3
    scale = 2.0 ** torch.randint(-2, 3, (N, K // 64), generator=g)
    W = (fp4[idx] * scale.repeat_interleave(64, dim=1)).to(torch.bfloat16)
    x = torch.randint(-4, 5, (M, K), generator=g).to(torch.bfloat16)
    q, st = F.quantize_4bit(W.to(DEV), blocksize=64, quant_type="fp4")
    with _forced(6000, K_TALL):
        y = _run_kernel(2, x.to(DEV), q, st, None)
    y_ref = (x.double() @ F.dequantize_4bit(q, st).double().cpu().t()).to(torch.bfloat16)
    assert torch.equal(y.float().cpu(), y_ref.float())


# ------------------------------------------------------------------------------------------ callers of dequantize_4bit
@pytest.mark.parametrize("quant_type", ["nf4", "fp4"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32], ids=["fp16", "bf16", "fp32"])
@pytest.mark.parametrize("dim,blocksize", [(64, 64), (4096, 64), (2560, 128), (192, 32), (8192, 4096)])
def test_dequantize_4bit_rows_matches_gathered_dequantize(quant_type, dtype, dim, blocksize):
    """The fused row-gather kernel is bit-identical to dequantizing the table and indexing it, for int32 and
    int64 indices, repeated indices and any index shape."""
    F = _F()
    rows = 300
    W = (torch.randn(rows, dim) * 0.2).to(dtype)
    q, st = F.quantize_4bit(W.to(DEV), blocksize=blocksize, quant_type=quant_type)
    table = O.dequantize_4bit(q.cpu(), st.absmax.cpu(), blocksize, quant_type, (rows, dim), dtype)
    for idx in (torch.tensor([0, rows - 1, 5, 5, 17]), torch.randint(0, rows, (3, 7)), torch.randint(0, rows, (1,))):
        for it in (torch.int64, torch.int32):
            out = torch.ops.bitsandbytes_amd.dequantize_4bit_rows.default(
                q, st.absmax, idx.to(DEV).to(it), dim, blocksize, quant_type, dtype)
            assert out.shape == (*idx.shape, dim) and out.dtype == dtype
            assert same_values_ftz(out.cpu(), table[idx])
    # out-of-range rows are NaN rows (documented: loud in the result, never an out-of-bounds read, never a plausible embedding)
    bad = torch.tensor([rows, -1, 2], device=DEV)
    out = torch.ops.bitsandbytes_amd.dequantize_4bit_rows.default(q, st.absmax, bad, dim, blocksize, quant_type, dtype)
    assert torch.isnan(out[:2]).all() and same_values_ftz(out[2].cpu(), table[2])


@pytest.mark.parametrize("cls_name,dim", [("EmbeddingNF4", 1024), ("EmbeddingFP4", 1024), ("EmbeddingNF4", 72),
                                          ("EmbeddingNF4", 96)])
def test_embedding4bit_gpu(cls_name, dim):
    """reference tests/test_modules.py::test_embedding_lossless-style check: lookups equal rows of the
    dequantized table (bit for bit) and are close to the fp table."""
    import bitsandbytes_amd.nn as bnn

    F = _F()
    torch.manual_seed(2)
    fp = torch.nn.Embedding(500, dim, dtype=torch.bfloat16)
    emb = getattr(bnn, cls_name)(500, dim, dtype=torch.bfloat16)
    emb.load_state_dict(fp.state_dict())
    emb = emb.to(DEV)
    assert emb.weight.dtype == torch.uint8 and emb.weight.bnb_quantized
    idx = torch.randint(0, 500, (4, 33), device=DEV)
    out = emb(idx)
    st = emb.weight.quant_state
    # the ORACLE's dequantized table (not this library's own dequantize kernel) is the reference of the lookup
    table = O.dequantize_4bit(emb.weight.data.cpu(), st.absmax.cpu(), st.blocksize, st.quant_type, (500, dim), torch.bfloat16)
    assert out.dtype == torch.bfloat16 and same_values_ftz(out.cpu(), table[idx.cpu()])
    assert torch.equal(out, F.dequantize_4bit(emb.weight.data, st)[idx])  # and the two device paths agree bit for bit
    assert rel_err(out.float().cpu(), fp(idx.cpu()).float()) < 0.25


def test_replace_parameter_4bit_gpu():
    import bitsandbytes_amd.nn.parametrize as bp

    F = _F()

    class Experts(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.randn(8, 256, 512, dtype=torch.bfloat16) * 0.05)

        def forward(self, x):
            return torch.einsum("bi,eoi->beo", x, self.w)

    m = Experts().to(DEV)
    w_fp = m.w.detach().clone()
    bp.replace_parameter_4bit(m, "w", compress_statistics=True, quant_type="nf4")
    q_o, am_o = O.quantize_4bit(w_fp.cpu(), 64, "nf4")
    assert torch.equal(m.parametrizations.w.original.cpu(), q_o)
    assert m.w.shape == w_fp.shape and m.w.dtype == torch.bfloat16
    assert rel_err(m.w.float().cpu(), w_fp.float().cpu()) < 0.12
    x = torch.randn(5, 512, device=DEV, dtype=torch.bfloat16)
    y = m(x)
    assert rel_err(y.float().cpu(), torch.einsum("bi,eoi->beo", x, w_fp).float().cpu()) < 0.12
    sd = m.state_dict()
    assert sd["w"].dtype == torch.uint8 and "w.quant_state.bitsandbytes__nf4" in sd and "w.nested_absmax" in sd


# ------------------------------------------------------------------------------------------ torch.library conformance
def _op_samples():
    F = _F()
    W = (torch.randn(64, 256, device=DEV) / 16).bfloat16()
    q, st = F.quantize_4bit(W, quant_type="nf4")
    qd, std = F.quantize_4bit(W, quant_type="fp4", blocksize=128, compress_statistics=True)
    x1 = torch.randn(1, 256, device=DEV, dtype=torch.bfloat16)
    x9 = torch.randn(9, 256, device=DEV, dtype=torch.bfloat16)
    bias = torch.randn(64, device=DEV, dtype=torch.bfloat16)
    code = F.create_dynamic_map().to(DEV)
    A8 = torch.randn(1024, device=DEV)
    q8, am8 = torch.ops.bitsandbytes.quantize_blockwise.default(A8, code, 256)
    ops = torch.ops.bitsandbytes
    return [
        (ops.quantize_4bit.default, (W, 64, "nf4", torch.uint8), {}),
        (ops.quantize_4bit.default, (W.float(), 128, "fp4", torch.bfloat16), {}),
        (ops.dequantize_4bit.default, (q, st.absmax, 64, "nf4", (64, 256), torch.bfloat16), {}),
        (ops.gemm_4bit.default, (x1, q, (64, 256), st.absmax, 64, "nf4"), {}),
        (ops.gemm_4bit.default, (x9, q, (64, 256), st.absmax, 64, "nf4", bias), {}),
        (ops.gemm_4bit.default, (x9, qd, (64, 256), std.state2.absmax, 128, "fp4", None, std.absmax, std.state2.code,
                                 std.offset), {}),
        (ops.gemv_4bit.default, (x1, q, (64, 256), st.absmax, st.code, 64), {}),
        (ops.quantize_blockwise.default, (A8, code, 256), {}),
        (ops.dequantize_blockwise.default, (q8, am8, code, 256, torch.float32), {}),
        (torch.ops.bitsandbytes_amd.dequantize_4bit_rows.default,
         (q, st.absmax, torch.tensor([3, 1, 3], device=DEV), 256, 64, "nf4", torch.bfloat16), {}),
        # fused backward: the fused kernel (whole tiles), its nested form, and the op's own unfused fallback (fp32 gradients)
        (torch.ops.bitsandbytes_amd.gemm_4bit_grad_input.default,
         (torch.randn(9, 64, device=DEV, dtype=torch.bfloat16), q, (64, 256), st.absmax, 64, "nf4"), {}),
        (torch.ops.bitsandbytes_amd.gemm_4bit_grad_input.default,
         (torch.randn(2, 3, 64, device=DEV, dtype=torch.bfloat16), qd, (64, 256), std.state2.absmax, 128, "fp4", std.absmax,
          std.state2.code, std.offset), {}),
        (torch.ops.bitsandbytes_amd.gemm_4bit_grad_input.default,
         (torch.randn(5, 64, device=DEV), q, (64, 256), st.absmax, 64, "nf4"), {}),
    ]


def test_opcheck_schema_and_fake_kernels():
    """torch.library.opcheck on every op of the path (reference tests/test_ops.py:150-232 does the same):
    schema correctness and fake-kernel/real-kernel agreement (shapes, dtypes, strides, devices)."""
    for op, args, kwargs in _op_samples():
        torch.library.opcheck(op, args, kwargs, test_utils=("test_schema", "test_faketensor"))


def test_native_dispatch_matches_python_kernel():
    """bitsandbytes::gemm_4bit is served by the C++-registered kernel (csrc/torch_dispatch.cpp); it must route, validate and
    compute exactly like the Python kernel it replaces: fused stream / MFMA shapes, nested absmax, bias, batch dims, the
    unfused large-batch route, the misaligned-K warning route, non-contiguous inputs, and the same errors."""
    import warnings

    from bitsandbytes_amd.backends import hip

    if os.environ.get("BNB_MI355X_PYTHON_DISPATCH") == "1":
        pytest.skip("the Python kernel was requested instead of the C++-registered one")
    assert hip.NATIVE_DISPATCH, "libbitsandbytes_mi355x_torch.so was not loaded"
    F = _F()
    op = torch.ops.bitsandbytes.gemm_4bit.default
    torch.manual_seed(11)
    for (lead, N, K, bs, qt, dq, dtype, with_bias) in [
        ((1,), 512, 1024, 64, "nf4", False, torch.bfloat16, False),
        ((2, 3), 384, 2048, 128, "fp4", True, torch.float16, True),
        ((24,), 256, 1024, 64, "nf4", True, torch.bfloat16, True),
        ((64,), 512, 2048, 64, "nf4", False, torch.bfloat16, False),
        ((3,), 256, 1024, 64, "nf4", False, torch.float32, True),
        ((8,), 256, 1024, 64, "fp4", False, torch.float32, False),       # fp32 above 4 rows: unfused
        ((200,), 256, 1024, 64, "nf4", True, torch.bfloat16, True),      # tall batch on the fused kernels (several row passes)
        ((600,), 256, 1024, 64, "nf4", True, torch.bfloat16, True),      # long rows (K >= 2 N): fused up to 1024 rows
        ((1100,), 256, 1024, 64, "nf4", True, torch.bfloat16, True),     # ... and unfused above (nested: ONE dequantize launch)
        ((700,), 1024, 1024, 64, "nf4", False, torch.float16, True),     # square: fused up to 640 rows, this one unfused
        ((12,), 96, 2752, 64, "nf4", True, torch.bfloat16, False),       # K % 256 != 0 on < 128 rows: the streaming kernel's passes up to 16 rows
        ((48,), 96, 2752, 64, "nf4", True, torch.bfloat16, True),        # ... above that dequantize + GEMM (round 5: was fused to 512)
        ((48,), 512, 2752, 64, "nf4", True, torch.bfloat16, True),       # ... from 128 rows on: the streaming MFMA kernel (round 6) ...
        ((100,), 3200, 1344, 64, "nf4", True, torch.bfloat16, True),     # ... its 32-row instances in row passes up to 128 rows
        ((140,), 3200, 1344, 64, "nf4", False, torch.bfloat16, False),   # ... and dequantize + GEMM above
        ((48,), 512, 2048, 32, "nf4", True, torch.bfloat16, False),      # blocksize 32 with nested statistics: likewise
        ((5,), 64, 96, 64, "nf4", False, torch.bfloat16, True),          # K % blocksize != 0: warning + unfused
    ]:
        W = (torch.randn(N, K, device=DEV) / K**0.5).to(dtype)
        q, st = F.quantize_4bit(W, blocksize=bs, quant_type=qt, compress_statistics=dq)
        x = torch.randn(*lead, K, device=DEV, dtype=dtype)
        bias = torch.randn(N, device=DEV, dtype=dtype) if with_bias else None
        if dq:
            args = (x, q, st.shape, st.state2.absmax, bs, qt, bias, st.absmax, st.state2.code, st.offset)
        else:
            args = (x, q, st.shape, st.absmax, bs, qt, bias)
        with warnings.catch_warnings(record=True) as w_native:
            warnings.simplefilter("always")
            got = op(*args)
        with warnings.catch_warnings(record=True) as w_py:
            warnings.simplefilter("always")
            want = hip._gemm_4bit_python_kernel(*args)
        assert got.shape == (*lead, N) and got.dtype == dtype
        assert torch.equal(got, want), (lead, N, K, bs, qt, dq, dtype)
        misaligned = K % bs != 0
        assert any("not aligned for fast kernel" in str(m.message) for m in w_native) == misaligned
        assert any("not aligned for fast kernel" in str(m.message) for m in w_py) == misaligned
    # non-contiguous activations are made contiguous by the glue
    xt = torch.randn(K, 4, device=DEV, dtype=dtype).t()
    assert torch.equal(op(xt, q, st.shape, st.absmax, bs, qt), hip._gemm_4bit_python_kernel(xt.contiguous(), q, st.shape, st.absmax, bs, qt))
    # the same argument errors
    W = torch.randn(64, 128, device=DEV).bfloat16()
    q, st = F.quantize_4bit(W, blocksize=64, quant_type="nf4")
    x = torch.randn(2, 128, device=DEV).bfloat16()
    for bad in (
        lambda f: f(x[:, :64], q, st.shape, st.absmax, 64, "nf4"),                                   # inner dim mismatch
        lambda f: f(x, q, st.shape, st.absmax.half(), 64, "nf4"),                                    # absmax dtype
        lambda f: f(x, q, st.shape, st.absmax, 64, "nf4", torch.zeros(64, device=DEV)),              # bias dtype
        lambda f: f(x, q, st.shape, st.absmax, 64, "nf4", torch.zeros(1, 64, device=DEV).bfloat16()),  # bias rank
    ):
        for f in (op, hip._gemm_4bit_python_kernel):
            with pytest.raises(RuntimeError):
                bad(f)


def test_linear4bit_under_torch_compile_aot_eager():
    """Fake kernels are sufficient to trace Linear4bit without graph breaks (reference
    tests/test_linear4bit.py:359-420 compiles with inductor; aot_eager needs no host C++ toolchain)."""
    import bitsandbytes_amd.nn as bnn

    torch.manual_seed(4)
    torch._dynamo.reset()
    net = torch.nn.Sequential(
        bnn.LinearNF4(256, 512, compute_dtype=torch.bfloat16), torch.nn.GELU(), bnn.LinearNF4(512, 256, compute_dtype=torch.bfloat16)
    ).to(DEV)
    x = torch.randn(3, 256, device=DEV, dtype=torch.bfloat16)
    want = net(x)
    compiled = torch.compile(net, backend="aot_eager", fullgraph=True)
    with torch.no_grad():
        got = compiled(x)
        got1 = compiled(x[:1])
    assert torch.equal(got, want)
    assert torch.equal(got1, net(x[:1]))


# ------------------------------------------------------------------------------------------ streaming dot kernel (round 2)
def _oracle_y_full(x, q, st, bias=None):
    """fp32-dequantize (oracle) + CPU fp64 matmul over ALL rows: the same oracle as `_oracle_y`, fast enough for the
    BASELINE-sized matrices (the scalar C gemm of the oracle is not)."""
    N, K = int(st.shape[0]), int(st.shape[1])
    if st.nested:
        absmax = O.dequantize_blockwise(st.absmax.cpu(), st.state2.absmax.cpu(), st.state2.code.cpu(), 256, torch.float32)
        absmax = absmax + st.offset.cpu().float()
    else:
        absmax = st.absmax.cpu()
    W = O.dequantize_4bit(q.cpu(), absmax, st.blocksize, st.quant_type, (N, K), torch.float32)
    y = x.cpu().double() @ W.double().t()
    if bias is not None:
        y = y + bias.cpu().double()
    return y


STREAM_SHAPES = [  # (M, N, K)
    (1, 4096, 4096), (1, 256, 2048), (1, 31, 96), (2, 300, 2816), (3, 130, 1024), (4, 512, 4096), (1, 16, 53248),
    (2, 64, 14336), (4, 40, 11008), (7, 96, 4096), (1, 1000, 32), (2, 5, 6144), (4, 8, 36864), (1, 300, 8192),
]


@pytest.mark.parametrize("M,N,K", STREAM_SHAPES)
@pytest.mark.parametrize("variant", ["bf16-nf4-64", "fp16-nf4-64", "fp32-nf4-64", "bf16-fp4-128-nested", "fp16-nf4-32",
                                     "bf16-nf4-4096", "fp32-fp4-64-nested", "bf16-nf4-64-nested"])
def test_stream_kernel_vs_oracle(M, N, K, variant):
    """csrc/gemv4_stream.hip through bnb_mi355x_gemm_4bit(kernel = 3): every activation dtype, both code tables,
    nested absmax, blocksizes from one lane's run (32) to two segments (4096), ragged K tails (K % 2048 != 0), rows
    longer than one workgroup's wavefronts (K = 36864 / 53248: several phases), M = 7 (passes of 4 over grid.y)."""
    F = _F()
    parts = variant.split("-")
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[parts[0]]
    qt, bs, dq = parts[1], int(parts[2]), "nested" in parts
    if K % bs:
        pytest.skip("K must be a multiple of the blocksize for the fused op")
    W = (torch.randn(N, K) / K**0.5).to(dt)
    x = torch.randn(M, K).to(dt)
    bias = torch.randn(N).to(dt)
    q, st = F.quantize_4bit(W.to(DEV), blocksize=bs, quant_type=qt, compress_statistics=dq)
    y_ref = _oracle_y_full(x, q, st, bias)
    y = _run_kernel(3, x.to(DEV), q, st, bias.to(DEV))
    assert y.shape == (M, N) and y.dtype == dt
    tol = 2e-5 if dt == torch.float32 else REL_TOL
    assert rel_err(y.cpu(), y_ref) < tol
    # the fp32 accumulation itself (before the output rounding) is much tighter than the tolerance
    if dt != torch.float32:
        assert rel_err(y.cpu(), y_ref.to(dt)) < 3e-3
    assert torch.equal(y, _run_kernel(3, x.to(DEV), q, st, bias.to(DEV)))  # bit-reproducible


@pytest.mark.parametrize("M,N,K", [(1, 4096, 4096), (2, 1376, 4096), (1, 512, 11008), (4, 300, 8192), (1, 64, 36864)])
def test_stream_kernel_geometry_independent(M, N, K):
    """Segment partials are combined in fixed order: ring depth, segments side by side, rows per workgroup, cache
    policy and wavefront count must not change a single bit (the reference demands run-to-run equality,
    tests/test_functional.py:1016-1034; here it also holds across launch geometries)."""
    import bitsandbytes_amd as bnb

    F = _F()
    W = (torch.randn(N, K) / K**0.5).bfloat16()
    x = torch.randn(M, K).bfloat16()
    q, st = F.quantize_4bit(W.to(DEV), blocksize=64, quant_type="nf4")
    y0 = _run_kernel(3, x.to(DEV), q, st)
    assert rel_err(y0.cpu(), _oracle_y_full(x, q, st)) < REL_TOL
    try:
        for tune in [(2, 0, 0, -1, 0), (3, 0, 0, -1, 0), (6, 0, 0, -1, 0), (0, 1, 0, -1, 0), (0, 0, 3, -1, 0),
                     (0, 0, 0, 0, 0), (0, 0, 0, -1, 8), (0, 2, 7, 0, 8)]:
            bnb.lib.bnb_mi355x_set_stream_tuning(*tune)
            assert torch.equal(_run_kernel(3, x.to(DEV), q, st), y0), f"tuning {tune}"
    finally:
        bnb.lib.bnb_mi355x_set_stream_tuning(0, 0, 0, -1, 0)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32], ids=["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("dq", [False, True], ids=["absmax32", "nested"])
@pytest.mark.parametrize("M", [1, 2, 4, 6])
def test_matmul_4bit_grouped_equals_separate_calls(dtype, dq, M):
    """bnb_mi355x_gemm_4bit_grouped: Q/K/V-like (4096 / 1024 / 1024 rows) and gate/up-like groups, with and without
    bias, must be bit-identical to one matmul_4bit per matrix (M = 6 exercises the library's own fallback)."""
    import bitsandbytes_amd as bnb

    F = _F()
    K = 2048
    x = torch.randn(M, K, device=DEV).to(dtype)
    for Ns, qt in (((1024, 256, 256), "nf4"), ((704, 704), "fp4"), ((96,), "nf4"), (tuple([64] * 9), "nf4")):
        ws, sts, bs_ = [], [], []
        for i, N in enumerate(Ns):
            W = (torch.randn(N, K, device=DEV) / K**0.5).to(dtype)
            q, st = F.quantize_4bit(W, blocksize=64, quant_type=qt, compress_statistics=dq)
            ws.append(q)
            sts.append(st)
            bs_.append(torch.randn(N, device=DEV).to(dtype) if i % 2 == 0 else None)
        ys = bnb.matmul_4bit_grouped(x, ws, sts, bs_)
        for y, w, s, b in zip(ys, ws, sts, bs_):
            y1 = bnb.matmul_4bit(x, w, s, bias=b)
            assert y.shape == y1.shape and torch.equal(y, y1)


@pytest.mark.parametrize("M", [2, 3, 4, 7, 8, 16])
def test_matmul_4bit_grouped_big_members_equal_separate_calls(M):
    """Two to sixteen rows: every member's own route is the streaming MFMA kernel, and the group is ONE launch of that kernel over
    the members' rows (round 6; until then a group with an MFMA-routed member was issued matrix by matrix) - bit-identical to
    separate calls, because the kernel's summation order does not depend on the number of tiles a workgroup holds."""
    import ctypes as ct

    import bitsandbytes_amd as bnb

    F = _F()
    K = 4096
    x = torch.randn(M, K, device=DEV).bfloat16()
    ws, sts = [], []
    for N in (4096, 1024, 1024):
        W = (torch.randn(N, K, device=DEV) / K**0.5).bfloat16()
        q, st = F.quantize_4bit(W, blocksize=64, quant_type="nf4")
        ws.append(q)
        sts.append(st)
    assert bnb.lib.bnb_mi355x_gemm_4bit_route(0, 2, M, 4096, K, 64) == 1 and bnb.lib.bnb_mi355x_gemm_4bit_route(0, 2, M, 1024, K, 64) == 1
    assert bnb.lib.bnb_mi355x_gemm_4bit_grouped_route(2, 3, (ct.c_int * 3)(4096, 1024, 1024), M, K, 64) == 2
    ys = bnb.matmul_4bit_grouped(x, ws, sts, [None] * 3)
    assert bnb.lib.bnb_mi355x_last_gemm_kernel() == K_SM
    for y, w, s in zip(ys, ws, sts):
        assert torch.equal(y, bnb.matmul_4bit(x, w, s))


@pytest.mark.parametrize("dtype,qt,bs,dq", [(torch.bfloat16, "nf4", 64, False), (torch.bfloat16, "nf4", 64, True), (torch.float16, "fp4", 128, True),
                                             (torch.float16, "nf4", 256, False)])
def test_grouped_launch_of_the_streaming_mfma_kernel(dtype, qt, bs, dq):
    """bnb_mi355x_gemm_4bit_grouped as one launch of the streaming MFMA kernel: groups of two to eight members of different heights
    (one to four tiles per workgroup, members that are not whole workgroup shares, a member of 144 rows), K with a partial last chunk,
    fp32 and nested statistics, bias on some members - against the oracle, bit-identical to the members' separate calls, the family
    that ran asserted; a group with a member below the kernel's range (64 rows) is issued matrix by matrix, same bits."""
    import ctypes as ct

    import bitsandbytes_amd as bnb

    F = _F()
    for (K, heights) in ((4096, (4096, 4096, 4096, 4096)), (2048, (11008, 11008)), (1024, (144, 3200, 528)), (2752, (1376, 1376)),
                         (512, (256,) * 8), (8192, (8192, 1024, 1024)), (4096, (4096, 64))):
        g = torch.Generator().manual_seed(K + len(heights))
        ws, sts, bs_ = [], [], []
        for i, N in enumerate(heights):
            W = (torch.randn(N, K, generator=g) / K**0.5).to(dtype)
            q, st = F.quantize_4bit(W.to(DEV), blocksize=bs, quant_type=qt, compress_statistics=dq)
            ws.append(q)
            sts.append(st)
            bs_.append(torch.randn(N, generator=g).to(dtype).to(DEV) if i % 2 else None)
        for M in (2, 5, 16):
            x = torch.randn(M, K, generator=g).to(dtype)
            want = bnb.lib.bnb_mi355x_gemm_4bit_grouped_route(1 if dtype == torch.float16 else 2, len(heights), (ct.c_int * len(heights))(*heights), M, K, bs)
            ys = bnb.matmul_4bit_grouped(x.to(DEV), ws, sts, bs_)
            if want == 2:
                assert bnb.lib.bnb_mi355x_last_gemm_kernel() == K_SM, (K, heights, M)
            if (heights == (4096,) * 4 or heights == (1376, 1376)) and K % bs == 0:
                assert want == 2, (K, heights, M)
            if min(heights) < 128:
                assert want != 2, (K, heights, M)
            for y, w, s, b in zip(ys, ws, sts, bs_):
                assert torch.equal(y, bnb.matmul_4bit(x.to(DEV), w, s, bias=b)), (K, heights, M)
                assert rel_err(y.cpu(), _oracle_y_full(x, w, s, b)) < REL_TOL, (K, heights, M)


@pytest.mark.parametrize("dq", [False, True], ids=["absmax32", "nested"])
def test_linear4bit_group_forward_prepared_call_equals_the_layers(dq):
    """linear4bit_group_forward on layers that hold a prepared call is ONE native call (csrc/torch_dispatch.cpp:
    linear4bit_group_prepared) and, where the library groups, one launch: results equal the layers called one by one - bit for bit up
    to 16 rows -, for bf16 and fp16 inputs, an fp32 input on bf16-compute layers (dtype policy of Linear4bit.forward), with and
    without bias, leading batch dimensions; a layer whose bias was replaced drops out of the prepared form and the Python path gives
    the same values."""
    import bitsandbytes_amd as bnb
    import bitsandbytes_amd.nn as bnn

    torch.manual_seed(11)
    K = 2048
    for dt in (torch.bfloat16, torch.float16):
        layers = [bnn.Linear4bit(K, n, bias=b, compute_dtype=dt, quant_type="nf4", compress_statistics=dq).to(DEV) for n, b in ((2048, True), (512, False), (512, True))]
        for shape, xdt in (((1, K), dt), ((3, K), dt), ((2, 5, K), dt), ((16, K), dt), ((4, K), torch.float32), ((40, K), dt)):
            x = torch.randn(*shape, device=DEV).to(xdt)
            with torch.no_grad():
                ref = [layer(x) for layer in layers]
                ref = [layer(x) for layer in layers]  # (second call: the prepared form)
                assert all(layer._prepared is not None for layer in layers)
                ys = bnn.linear4bit_group_forward(layers, x)
            M = x.numel() // K
            for y, r in zip(ys, ref):
                assert y.shape == r.shape and y.dtype == r.dtype
                if M <= 16:
                    assert torch.equal(y, r), (dt, shape, xdt)
                else:
                    assert rel_err(y.float().cpu(), r.float().cpu()) < REL_TOL
        # under hipGraph capture (no allocation outside torch's allocator, no synchronisation): replays compute on new inputs
        xg = torch.randn(4, K, device=DEV).to(dt)
        with torch.no_grad():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                bnn.linear4bit_group_forward(layers, xg)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                yg = bnn.linear4bit_group_forward(layers, xg)
            for _ in range(3):
                xg.copy_(torch.randn(4, K, device=DEV).to(dt))
                graph.replay()
                torch.cuda.synchronize()
                for y, layer in zip(yg, layers):
                    assert torch.equal(y, layer(xg))
        with torch.no_grad():
            layers[0].bias = torch.nn.Parameter(torch.randn(2048, device=DEV).to(dt))
            x = torch.randn(2, K, device=DEV).to(dt)
            ys = bnn.linear4bit_group_forward(layers, x)
            for y, layer in zip(ys, layers):
                assert torch.equal(y, layer(x))


@pytest.mark.parametrize("M", [17, 24, 32, 33, 48, 64])
def test_grouped_launch_in_row_passes_from_17_rows(M):
    """Groups of 17 ... 64 rows up to a measured size (csrc/c_api.hip: grouped_sm_passes) are ONE launch of the streaming MFMA kernel
    in row passes of 16 - the members' own route at that many rows is another MFMA kernel, so this is the one grouped form that is
    NOT bit-identical to separate calls: values against the oracle (per row of the batch), the family that ran, bit-reproducible run
    to run; a group beyond the size keeps the members' own kernels."""
    import ctypes as ct

    import bitsandbytes_amd as bnb

    F = _F()
    for (K, heights, dq) in ((4096, (4096, 1024, 1024), False), (4096, (512, 512, 512), True), (2752, (1376, 1376), False), (4096, (11008, 11008), False)):
        g = torch.Generator().manual_seed(K + len(heights) + M)
        ws, sts, bs_ = [], [], []
        for i, N in enumerate(heights):
            W = (torch.randn(N, K, generator=g) / K**0.5).to(torch.bfloat16)
            q, st = F.quantize_4bit(W.to(DEV), blocksize=64, quant_type="nf4", compress_statistics=dq)
            ws.append(q)
            sts.append(st)
            bs_.append(torch.randn(N, generator=g).to(torch.bfloat16).to(DEV) if i % 2 else None)
        x = torch.randn(M, K, generator=g).to(torch.bfloat16)
        want = bnb.lib.bnb_mi355x_gemm_4bit_grouped_route(2, len(heights), (ct.c_int * len(heights))(*heights), M, K, 64)
        weights = sum(heights) * K
        assert (want == 2) == (weights <= ((96 << 20) if M <= 32 else (72 << 20))), (K, heights, M, want)
        ys = bnb.matmul_4bit_grouped(x.to(DEV), ws, sts, bs_)
        if want == 2:
            assert bnb.lib.bnb_mi355x_last_gemm_kernel() == K_SM, (K, heights, M)
            ys2 = bnb.matmul_4bit_grouped(x.to(DEV), ws, sts, bs_)
        for i, (y, w, s, b) in enumerate(zip(ys, ws, sts, bs_)):
            y_ref = _oracle_y_full(x, w, s, b)
            assert rel_err(y.cpu(), y_ref) < REL_TOL, (K, heights, M)
            worst = ((y.double().cpu() - y_ref.double()).norm(dim=1) / y_ref.double().norm(dim=1)).max()
            assert worst < 2 * REL_TOL, (K, heights, M, float(worst))
            if want == 2:
                assert torch.equal(y, ys2[i])
            else:
                assert torch.equal(y, bnb.matmul_4bit(x.to(DEV), w, s, bias=b))


def test_mfma_sm_kernel_rows_do_not_depend_on_the_launch_geometry():
    """A weight row's result is the same bits whether its matrix gives a workgroup one tile (16 wavefronts) or four (8 wavefronts, two
    accumulator sets per tile = the sixteen wavefronts' chunk lists): rows [a, b) of a tall matrix equal the result of the matrix
    made of those rows alone - what row shards (parallel.py) and grouped launches rely on. 2 ... 8 rows (sixteen staged rows run 8
    wavefronts in every instance), K of one, several and a partial chunk per wavefront."""
    F = _F()
    for (N, K, cut) in ((16384, 4096, (4096, 5120)), (12288, 8192, (0, 2048)), (20000, 1344, (16000, 17024)), (11008, 2048, (5504, 8256))):
        from bitsandbytes_amd.backends import hip

        g = torch.Generator().manual_seed(N + K)
        W = (torch.randn(N, K, generator=g) / K**0.5).to(torch.bfloat16).to(DEV)
        q, st = F.quantize_4bit(W, blocksize=64, quant_type="nf4")
        a, b = cut
        qs = q.view(N, K // 2)[a:b].contiguous().view(-1, 1)
        absmax_s = st.absmax.view(N, K // 64)[a:b].contiguous().view(-1)
        for M in (2, 3, 4, 7, 8):
            x = torch.randn(M, K, generator=g).to(torch.bfloat16).to(DEV)
            with _forced(5000, K_SM):
                y_all = hip._gemm_4bit_fused(x, q, (N, K), st.absmax, 64, "nf4", None, None, None, None, kernel=2)
                y_cut = hip._gemm_4bit_fused(x, qs, (b - a, K), absmax_s, 64, "nf4", None, None, None, None, kernel=2)
            assert torch.equal(y_all[:, a:b], y_cut), (N, K, cut, M)


def test_matmul_4bit_double_backward_gpu():
    """create_graph=True through matmul_4bit: the backward must stay differentiable in grad_output (reference formulation)."""
    import bitsandbytes_amd as bnb

    F = _F()
    W = (torch.randn(256, 512, device=DEV) / 512**0.5).bfloat16()
    q, st = F.quantize_4bit(W, blocksize=64, quant_type="nf4")
    x = torch.randn(8, 512, device=DEV).bfloat16().requires_grad_(True)
    y = bnb.matmul_4bit(x, q, st)
    (gx,) = torch.autograd.grad(y.float().pow(2).sum(), x, create_graph=True)
    assert gx.requires_grad
    gx.float().pow(2).sum().backward()       # second derivative reaches x through grad_output
    assert x.grad is not None and torch.isfinite(x.grad.float()).all()


def test_gemv_fp32_summation_is_as_accurate_as_the_blas_library():
    """The reference's fp32 gemv envelope compares gemv_4bit with the device BLAS's F.linear over the dequantized weight (two
    fp32 summation orders). Here both are compared with the EXACT result (fp64 over the same dequantized values) on the "fc2"
    shapes where that envelope is missed on rocBLAS: the fused kernel's error must not exceed the BLAS library's by more than
    a quarter - the deviation is a difference between two equally accurate orders, not a loss of accuracy."""
    F = _F()
    torch.manual_seed(3)
    for dim in (128, 1024):
        ours, blas = [], []
        for _ in range(10):
            A = torch.randn(1, 4 * dim, device=DEV)
            B = torch.randn(dim, 4 * dim, device=DEV) / dim**0.5
            q, st = F.quantize_4bit(B, quant_type="nf4")
            Wd = F.dequantize_4bit(q, st)
            exact = A.double() @ Wd.double().t()
            c2 = F.gemv_4bit(A, q.t(), state=st)
            c1 = torch.nn.functional.linear(A, Wd)
            ours.append((c2.double() - exact).abs().mean().item())
            blas.append((c1.double() - exact).abs().mean().item())
        assert sum(ours) <= 1.25 * sum(blas) + 1e-12, (dim, sum(ours) / 10, sum(blas) / 10)


def test_linear4bit_prepared_call_equals_ordinary_path():
    """Linear4bit.forward's prepared call (csrc/torch_dispatch.cpp linear4bit_prepare / linear4bit_prepared): from the second
    eager no-grad call on, the layer runs a two-argument C++ op instead of the Python layers. Same results bit for bit - bias,
    nested statistics, fp32 input with bf16 compute, several batch sizes incl. the unfused range; the handle is dropped
    when the weight, the quant state or the bias object changes, and autograd calls keep the ordinary path."""
    from bitsandbytes_amd.backends import hip
    from bitsandbytes_amd.nn import Linear4bit

    if not hip.NATIVE_DISPATCH:
        pytest.skip("the C++ dispatcher library is not loaded")
    torch.manual_seed(11)
    for (K, N, cs, qt, cdt, xdt, with_bias) in ((1024, 512, True, "nf4", torch.bfloat16, torch.bfloat16, True),
                                                (2048, 256, False, "fp4", torch.float16, torch.float16, False),
                                                (1024, 384, True, "nf4", torch.bfloat16, torch.float32, True)):
        layer = Linear4bit(K, N, bias=with_bias, quant_type=qt, compress_statistics=cs, compute_dtype=cdt).to(DEV)
        for M in (1, 3, 40, 700):
            x = torch.randn(M, K, device=DEV, dtype=xdt)
            with torch.no_grad():
                layer._prepared_drop()
                y0 = layer(x)                       # ordinary path (prepares)
                assert layer._prepared is not None
                y1 = layer(x)                       # prepared call
                y2 = layer(x)
            assert y1.dtype == xdt and torch.equal(y0, y1) and torch.equal(y1, y2)
        # autograd keeps the ordinary path and still works
        xg = torch.randn(4, K, device=DEV, dtype=xdt, requires_grad=True)
        layer(xg).float().sum().backward()
        assert xg.grad is not None
        # a replaced bias / re-quantised weight drops the stale handle
        with torch.no_grad():
            h0 = layer._prepared.handle
            if with_bias:
                layer.bias = torch.nn.Parameter(torch.randn(N, device=DEV, dtype=cdt), requires_grad=False)
                x = torch.randn(2, K, device=DEV, dtype=xdt)
                ya = layer(x)                       # ordinary path again (bias object changed)
                yb = layer(x)
                assert layer._prepared.handle != h0 and torch.equal(ya, yb)


def test_linear4bit_prepared_call_never_serves_a_stale_bias_or_a_foreign_handle():
    """Round-3 review items: (a) `layer.bias.data = new` keeps the Parameter object - the prepared call must not keep serving the
    old values (it aliases the live storage and is keyed on its data_ptr), nor after an in-place update; (b) the handle is an
    owning object that never travels: deepcopy / pickle of a module carry no handle, and collecting the copy leaves the
    original's call intact; (c) `.to()` drops the handle (it holds the packed weight alive on the device)."""
    import copy
    import gc
    import io

    from bitsandbytes_amd.backends import hip
    from bitsandbytes_amd.nn import Linear4bit

    if not hip.NATIVE_DISPATCH:
        pytest.skip("the C++ dispatcher library is not loaded")
    torch.manual_seed(12)
    K, N = 1024, 256
    for (bias_dtype, cdt) in ((torch.bfloat16, torch.bfloat16), (torch.float32, torch.bfloat16)):
        layer = Linear4bit(K, N, bias=True, quant_type="nf4", compute_dtype=cdt).to(DEV)
        layer.bias.data = layer.bias.data.to(bias_dtype)
        x = torch.randn(1, K, device=DEV, dtype=torch.bfloat16)
        with torch.no_grad():
            y0 = layer(x)
            y1 = layer(x)
            assert layer._prepared is not None and torch.equal(y0, y1)
            # (a) swapped storage, same Parameter object
            new_bias = torch.randn(N, device=DEV, dtype=layer.bias.dtype)
            layer.bias.data = new_bias
            ya = layer(x)
            want = (y0.float() - (y0.float() * 0 + 0)).clone()  # placeholder shape
            layer._prepared_drop()
            yb = layer(x)                              # ordinary path on the same state
            assert torch.equal(ya, yb), "prepared call served a stale bias after `bias.data = new`"
            assert not torch.equal(ya, y0)
            # in-place update of the live storage
            layer(x)
            assert layer._prepared is not None
            layer.bias.data.add_(1.0)
            yc = layer(x)
            layer._prepared_drop()
            yd = layer(x)
            assert torch.equal(yc, yd), "prepared call served a stale bias after an in-place update"
            del want
            # (b) deepcopy / pickle carry no handle; collecting the copy leaves the original's handle valid
            layer(x)
            h = layer._prepared.handle
            twin = copy.deepcopy(layer)
            assert twin.__dict__.get("_prepared") is None
            yt = twin(x)
            yt2 = twin(x)
            assert torch.equal(yt, yd) and torch.equal(yt2, yd)
            assert twin._prepared is None or twin._prepared.handle != h
            del twin
            gc.collect()
            assert torch.equal(layer(x), yd) and layer._prepared.handle == h
            buf = io.BytesIO()
            torch.save(layer, buf)
            buf.seek(0)
            loaded = torch.load(buf, weights_only=False)
            assert loaded.__dict__.get("_prepared") is None
            assert torch.equal(loaded(x), yd)
            del loaded
            gc.collect()
            assert torch.equal(layer(x), yd)
            # (c) a move drops the handle
            layer.to(DEV)
            assert layer.__dict__.get("_prepared") is None
            assert torch.equal(layer(x), yd)


def test_linear4bit_group_forward_gpu():
    from bitsandbytes_amd.nn import Linear4bit, linear4bit_group_forward

    torch.manual_seed(5)
    layers = []
    for n_out in (512, 128, 128):
        ref = torch.nn.Linear(1024, n_out, bias=True)
        layer = Linear4bit(1024, n_out, bias=True, quant_type="nf4", compute_dtype=torch.bfloat16)
        layer.load_state_dict(ref.state_dict())
        layers.append(layer.to(DEV))
    for shape in ((1, 1024), (1, 3, 1024), (40, 1024)):
        x = torch.randn(*shape, device=DEV)
        ys = linear4bit_group_forward(layers, x)
        for y, layer in zip(ys, layers):
            assert torch.equal(y, layer(x))


# ------------------------------------------------------------------------------------------ BASELINE configs at full size
C4_SHAPES = [(11008, 4096), (4096, 11008), (1376, 4096), (512, 11008)]  # Llama FFN matrices and their 8-way row shards


@pytest.mark.parametrize("N,K", C4_SHAPES)
@pytest.mark.parametrize("M", [1, 64])
@pytest.mark.parametrize("dq", [False, True], ids=["absmax32", "nested"])
def test_config4_llama_ffn_shapes_full(N, K, M, dq):
    """BASELINE.json configs[3]: Linear4bit NF4 on the Llama-3-8B FFN shapes and the per-GPU shards of the 8-way
    row split (1376 = 10.75 x 128 columns is ragged against every MFMA tile), M = 1 (streaming kernel) and M = 64
    (MFMA producer/consumer kernel), every output row against the fp32-dequantize + fp64-matmul oracle."""
    import bitsandbytes_amd as bnb

    F = _F()
    W = (torch.randn(N, K, device=DEV) / K**0.5).bfloat16()
    x = torch.randn(M, K, device=DEV).bfloat16()
    bias = torch.randn(N, device=DEV).bfloat16()
    q, st = F.quantize_4bit(W, blocksize=64, quant_type="nf4", compress_statistics=dq)
    del W
    y = bnb.matmul_4bit(x, q, st, bias=bias)
    assert y.shape == (M, N)
    assert rel_err(y.cpu(), _oracle_y_full(x, q, st, bias)) < REL_TOL
    assert torch.equal(bnb.matmul_4bit(x, q, st, bias=bias), y)


@pytest.mark.parametrize("M", [1, 64])
def test_config3_8192_all_rows(M):
    """BASELINE.json configs[2] (gemm_4bit NF4 bf16 M = 64, N = K = 8192) checked on ALL 8192 output rows (and the
    same matrix at M = 1)."""
    import bitsandbytes_amd as bnb

    F = _F()
    N = K = 8192
    W = (torch.randn(N, K, device=DEV) / K**0.5).bfloat16()
    x = torch.randn(M, K, device=DEV).bfloat16()
    q, st = F.quantize_4bit(W, blocksize=64, quant_type="nf4")
    del W
    y = bnb.matmul_4bit(x, q, st)
    assert rel_err(y.cpu(), _oracle_y_full(x, q, st)) < REL_TOL


def test_config5_fp4_nested_bs128_full():
    """BASELINE.json configs[4]: FP4, double quant, blocksize 128, bf16, M = 1, N = K = 4096, all rows."""
    import bitsandbytes_amd as bnb

    F = _F()
    N = K = 4096
    W = (torch.randn(N, K, device=DEV) / K**0.5).bfloat16()
    x = torch.randn(1, K, device=DEV).bfloat16()
    q, st = F.quantize_4bit(W, blocksize=128, quant_type="fp4", compress_statistics=True)
    y = bnb.matmul_4bit(x, q, st)
    assert rel_err(y.cpu(), _oracle_y_full(x, q, st)) < REL_TOL


# ------------------------------------------------------------------------------------------ multi-GPU path on one GPU
@pytest.mark.parametrize("N,K", [(11008, 4096), (4096, 11008)])
@pytest.mark.parametrize("M", [1, 2, 64])
def test_config4_eight_row_shards_equal_the_full_layer(N, K, M):
    """BASELINE.json configs[3] as the sharded path computes it: the 8 row shards parallel.shard_linear4bit hands to the 8
    ranks (1376 x 4096, 512 x 11008; nested absmax, so the second-level slicing is exercised), each run on this GPU,
    concatenated = the un-sharded layer. Bit for bit at M = 1 (the streaming kernel's result does not depend on the launch
    geometry; round 6: from two rows on the full layer takes the streaming MFMA kernel, its narrow shards do not), within the
    matmul tolerance of the oracle above (the MFMA geometry follows N)."""
    import bitsandbytes_amd.nn as bnn
    from bitsandbytes_amd.parallel import shard_linear4bit

    torch.manual_seed(3)
    layer = bnn.Linear4bit(K, N, bias=True, compute_dtype=torch.bfloat16, compress_statistics=True, quant_type="nf4").to(DEV)
    x = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
    y_full = layer(x)
    parts = [shard_linear4bit(layer, rank=r, world_size=8, gather_output=False)(x) for r in range(8)]
    y_cat = torch.cat(parts, dim=-1)
    assert y_cat.shape == y_full.shape
    if M == 1:
        assert torch.equal(y_cat, y_full)
    else:
        assert rel_err(y_cat.cpu(), y_full.cpu()) < 3e-3


def test_sharded_linear4bit_over_rccl_world_size_one():
    """parallel.ShardedLinear4bit with torch.distributed backend "nccl" (= RCCL on ROCm) in a group of one: process-group
    initialisation on the device, the kernel on the shard, and the all_gather_into_tensor collective actually issued
    (always_gather) - the single-GPU rehearsal of BASELINE.json configs[3]; the 2-rank exchange is covered on CPU (gloo)."""
    import socket

    import torch.distributed as dist

    import bitsandbytes_amd.nn as bnn
    from bitsandbytes_amd.parallel import shard_linear4bit

    if dist.is_initialized():
        pytest.skip("a process group already exists in this interpreter")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    try:
        torch.manual_seed(4)
        for (N, K, M) in ((1376, 4096, 1), (512, 11008, 1), (1376, 4096, 64)):
            layer = bnn.Linear4bit(K, N, bias=True, compute_dtype=torch.bfloat16, quant_type="nf4").to(DEV)
            x = torch.randn(M, K, device=DEV, dtype=torch.bfloat16)
            sharded = shard_linear4bit(layer, always_gather=True)
            y = sharded(x)
            torch.cuda.synchronize()
            assert y.shape == (M, N)
            assert torch.equal(y, layer(x))
    finally:
        dist.destroy_process_group()


def test_bench_multi_gpu_code_path_at_world_size_one():
    """bench.py's --gpus N > 1 branch (the N-sharded MLP chain: ShardedLinear4bit shards, the fused peer chain checked against the
    separate-gather form at start-up, the other forms timed beside it) cannot run with N > 1 on a 1-GPU box; `--sharded-path` runs
    the same code with an RCCL group of one rank. The JSON line must be the LAST line of stdout (RCCL prints a banner through C
    stdio) and carry the contract's keys."""
    import json
    import subprocess
    import sys

    from conftest import ROOT

    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--sharded-path", "--steps", "6", "--warmup", "2"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 6 and line["value"] > 0
    sc = line["sharded_chain"]
    assert sc["H"] == 4096 and sc["F"] == 4096 and sc["up_shard"] == [4096, 4096]  # world 1: the headline layer itself
    assert sc["us_per_layer_kernels_alone"] > 0 and sc["us_per_layer_kernel_plus_separate_gather"] > 0
    # the gather is fused into the gemv launches wherever the peer chain constructs and reproduces the separate form (it must, here)
    assert "fused into the gemv launches" in sc["gather"] and sc["fused_chain_not_used_because"] is None, (sc, r.stderr[-2000:])
    assert "sharded x1" in line["config"]["parallelism"]


def test_sharded_linear4bit_over_rccl_two_ranks():
    """BASELINE.json configs[3] on real links: two ranks (one process per GPU, backend "nccl" = RCCL), each with its row shard of
    the Llama FFN matrices; the gathered output of every rank must equal the unsharded layer bit for bit (M = 1 and M = 2: the
    streaming kernel's results do not depend on the launch geometry). Self-skips on a box with fewer than two GPUs - the
    builder's boxes have one; the driver's multi-GPU node runs it."""
    import subprocess
    import sys
    import textwrap

    from conftest import ROOT

    if torch.cuda.device_count() < 2:
        pytest.skip(f"needs 2 GPUs, this box has {torch.cuda.device_count()}")
    script = textwrap.dedent(f"""
        import os, sys, torch, torch.distributed as dist
        sys.path.insert(0, {ROOT!r})
        import bitsandbytes_amd.nn as bnn
        from bitsandbytes_amd.parallel import shard_linear4bit
        rank = int(os.environ["RANK"]); torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        dist.init_process_group("nccl", device_id=dev)
        torch.manual_seed(7)                                   # the same layer on every rank
        for (N, K) in ((11008, 4096), (4096, 11008)):
            layer = bnn.Linear4bit(K, N, bias=True, compute_dtype=torch.bfloat16, quant_type="nf4").to(dev)
            sh = shard_linear4bit(layer)
            for M in (1, 2, 64):
                x = torch.randn(M, K, generator=torch.Generator().manual_seed(M)).to(dev, torch.bfloat16)
                y, y0 = sh(x), layer(x)
                assert y.shape == y0.shape
                if M == 1:
                    assert torch.equal(y, y0), (N, K, M)
                else:
                    assert ((y.float() - y0.float()).abs().max() / y0.float().abs().max()).item() < 1e-2
        dist.barrier(); dist.destroy_process_group()
        print("RANK_OK", rank)
    """)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29731", os.path.join(ROOT, "tests", "checks", "two_rank_nccl.py")],
                       capture_output=True, text=True, timeout=900, env=dict(env, BNB_TWO_RANK_SCRIPT=script))
    assert r.returncode == 0 and r.stdout.count("RANK_OK") == 2, r.stdout[-2000:] + r.stderr[-4000:]
