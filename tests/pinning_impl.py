"""(Run through tests/test_oracle_pinning.py, in its own process: the reference package and
bitsandbytes_amd both define the ``bitsandbytes::`` operator schemas, and the reference must be the
first to do so.)

CPU, authoring container only: the oracle against the LIVE reference (Python package imported from
/root/reference + its own libbitsandbytes_cpu.so from oracle/build_ref.sh) on fresh random inputs,
and against that library's C ABI directly. Skipped where the reference checkout is absent."""
import pytest
import torch

from conftest import same_values, same_values_ftz
from oracle import oracle as O
from oracle.ref_import import reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason="needs /root/reference and oracle/_ref (build_ref.sh)")


@pytest.fixture(scope="module")
def ref():
    from oracle.ref_import import import_reference

    bnb = import_reference()
    import bitsandbytes.functional as F

    return bnb, F


@pytest.mark.parametrize("quant_type", ["nf4", "fp4"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16], ids=["fp32", "fp16", "bf16"])
@pytest.mark.parametrize("blocksize", [32, 64, 256, 4096])
def test_quantize_4bit_live(ref, quant_type, dtype, blocksize):
    _, F = ref
    for n in (blocksize * 6, blocksize * 2 + 11, 4097):
        A = (torch.randn(n) * 0.7).to(dtype)
        A[::13] = 0
        q_ref, st = F.quantize_4bit(A, blocksize=blocksize, quant_type=quant_type)
        q, am = O.quantize_4bit(A, blocksize, quant_type)
        assert torch.equal(q, q_ref) and torch.equal(am, st.absmax)
        for odt in (torch.float32, torch.float16, torch.bfloat16):
            d_ref = torch.ops.bitsandbytes.dequantize_4bit.default(q_ref, st.absmax, blocksize, quant_type, (n,), odt)
            d = O.dequantize_4bit(q, am, blocksize, quant_type, (n,), odt)
            assert same_values_ftz(d, d_ref.reshape(-1))


def test_quantize_4bit_c1_full_size(ref):
    """BASELINE config 1: NF4 bs=64 on a 4096x4096 fp16 weight — all 16.7M codes and 262144 absmax."""
    _, F = ref
    W = torch.randn(4096, 4096).half()
    q_ref, st = F.quantize_4bit(W, blocksize=64, quant_type="nf4")
    q, am = O.quantize_4bit(W, 64, "nf4")
    assert torch.equal(q, q_ref) and torch.equal(am, st.absmax)
    d = O.dequantize_4bit(q, am, 64, "nf4", W.shape, torch.float16)
    assert same_values(d, F.dequantize_4bit(q_ref, st))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16], ids=["fp32", "fp16", "bf16"])
def test_blockwise_8bit_live_and_c_abi(ref, dtype):
    _, F = ref
    code = F.create_dynamic_map().float()
    A = torch.randn(256 * 37 + 5).to(dtype)
    A[256:512] = 0
    q_ref, am_ref = torch.ops.bitsandbytes.quantize_blockwise.default(A, code, 256)
    q, am = O.quantize_blockwise(A, code, 256)
    assert torch.equal(q, q_ref) and torch.equal(am, am_ref)
    # the reference library called straight through its C ABI (what bench.py's cpu_baseline uses)
    q_c, am_c = O.ref_quantize_blockwise(A, code, 256)
    assert torch.equal(q, q_c) and torch.equal(am, am_c)
    d = O.dequantize_blockwise(q, am, code, 256, torch.float32)
    assert torch.equal(d, O.ref_dequantize_blockwise(q, am, code, 256, torch.float32))


# Code tensors stay alive for the whole process. The reference's CPU kernel caches its 64K-entry table per code
# POINTER, guarded only by a 4-value fingerprint (entries 0, 1, 127, 255: csrc/cpu_ops.cpp:522-558); two different
# maps can share it - linear 4-bit and fp8 e5m2 both read (-1, -6/7, 0, 1) - so a freed code whose address is
# recycled for the other map makes the REFERENCE quantize with a stale table. Distinct live addresses avoid that
# (the oracle and the HIP kernel keep no cache).
_LIVE_CODES = []


@pytest.mark.parametrize("which", ["dynamic_unsigned", "linear8", "linear4", "fp8_e4m3", "fp8_e5m2", "fp4_as_map", "normal"])
def test_blockwise_8bit_other_code_maps_live(ref, which):
    """The oracle's 8-bit rule on code maps with repeated entries (few-bit maps are zero-padded to 256) and on
    non-dynamic 8-bit maps, against the reference op and the reference library's C ABI."""
    _, F = ref
    code = {
        "dynamic_unsigned": lambda: F.create_dynamic_map(signed=False),
        "linear8": lambda: F.create_linear_map(True, 8),
        "linear4": lambda: F.create_linear_map(True, 4),
        "fp8_e4m3": lambda: F.create_fp8_map(True, 4, 3, 8),
        "fp8_e5m2": lambda: F.create_fp8_map(True, 5, 2, 8),
        "fp4_as_map": lambda: F.create_fp8_map(True, 2, 1, 4),
        "normal": lambda: F.create_normal_map(),
    }[which]().float()
    _LIVE_CODES.append(code)
    g = torch.Generator().manual_seed(11)
    A = torch.randn(256 * 41 + 9, generator=g)
    if which == "dynamic_unsigned":
        A = A.abs()
    A[512:768] = 0
    A[::13] = 0
    for bs in (256, 64, 4096):
        q_ref, am_ref = torch.ops.bitsandbytes.quantize_blockwise.default(A, code, bs)
        q, am = O.quantize_blockwise(A, code, bs)
        assert torch.equal(q, q_ref) and torch.equal(am, am_ref), f"bs={bs}"
        q_c, am_c = O.ref_quantize_blockwise(A, code, bs)
        assert torch.equal(q, q_c) and torch.equal(am, am_c), f"bs={bs} (C ABI)"
        d = O.dequantize_blockwise(q, am, code, bs, torch.float32)
        assert torch.equal(d, O.ref_dequantize_blockwise(q, am, code, bs, torch.float32))
        assert torch.equal(d, torch.ops.bitsandbytes.dequantize_blockwise.default(q_ref, am_ref, code, bs, torch.float32))


def test_ref_c_abi_dequantize_4bit(ref):
    A = torch.randn(64, 256).bfloat16()
    q, am = O.quantize_4bit(A, 64, "nf4")
    for odt in (torch.float32, torch.bfloat16, torch.float16):
        assert same_values_ftz(O.ref_dequantize_4bit(q, am, 64, "nf4", (64, 256), odt),
                               O.dequantize_4bit(q, am, 64, "nf4", (64, 256), odt))


def test_dynamic_map_and_code_tables_match_reference(ref):
    _, F = ref
    import bitsandbytes_amd.functional as MF

    assert torch.equal(MF.create_dynamic_map().view(torch.int32), F.create_dynamic_map().float().view(torch.int32))
    for qt in ("nf4", "fp4"):
        assert torch.equal(MF.get_4bit_type(qt, device="cpu").view(torch.int32),
                           F.get_4bit_type(qt, device="cpu").view(torch.int32))


def test_every_code_map_constructor_matches_reference(ref):
    """create_dynamic_map / create_linear_map / create_fp8_map / create_normal_map: bit-identical tensors for the
    parameterisations the reference's own tests use (tests/test_functional.py:225-300) and then some."""
    _, F = ref
    import bitsandbytes_amd.functional as MF

    def same(a, b, what):
        assert a.dtype == b.dtype == torch.float32 and a.shape == b.shape == (256,), what
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), what

    def same_call(name, *args, **kw):
        """Equal tensors, or - for parameter combinations the reference rejects - the same rejection."""
        try:
            expect = getattr(F, name)(*args, **kw)
        except AssertionError:
            with pytest.raises(AssertionError):
                getattr(MF, name)(*args, **kw)
            return
        same(getattr(MF, name)(*args, **kw), expect, f"{name}{args}{kw}")

    for signed in (True, False):
        for bits in range(2, 9):
            same_call("create_linear_map", signed, total_bits=bits)
            for mx in range(1, bits + 1):
                same_call("create_dynamic_map", signed, mx, bits)
            for e in range(1, bits - (1 if signed else 0) + 1):
                same_call("create_fp8_map", signed, e, bits - e - (1 if signed else 0), bits)
        same_call("create_linear_map", signed, add_zero=False)
    same_call("create_fp8_map", True, 4, 4, 8)  # exponent + precision bits that do not add up: both reject
    for extra in (True, False):
        same_call("create_normal_map", use_extra_value=extra)
    same_call("create_normal_map", offset=0.95)


def test_quant_state_dict_format_matches_reference(ref):
    """Packed state-dict blob written by our QuantState is byte-identical to the reference's."""
    _, F = ref
    import bitsandbytes_amd.functional as MF

    W = torch.randn(32, 128).bfloat16()
    for dq in (False, True):
        q, st = F.quantize_4bit(W, blocksize=64, quant_type="nf4", compress_statistics=dq)
        ref_dict = st.as_dict(packed=True)
        mine = MF.QuantState.from_dict({k: v.clone() if isinstance(v, torch.Tensor) else v
                                        for k, v in st.as_dict(packed=True).items()}, device="cpu")
        my_dict = mine.as_dict(packed=True)
        assert set(ref_dict) == set(my_dict)
        for k in ref_dict:
            assert torch.equal(ref_dict[k], my_dict[k]), k


def test_ref_fused_cpu_gemv_baseline_path(ref):
    """The CPU-baseline path bench.py times (reference AVX512-BF16 fused gemv through its C ABI with
    our restatement of the reference's weight repack) computes the right thing."""
    if not O.ref_has_avx512bf16():
        pytest.skip("host lacks AVX512-BF16")
    bnb, F = ref
    N, K = 256, 512
    W = (torch.randn(N, K) / K**0.5).bfloat16()
    q, am = O.quantize_4bit(W, 64, "nf4")
    wp, amt = O.ref_pack_for_cpu_gemv(q, am, N, K, 64)
    # same packing as the reference's own Python repack
    q_ref, st = F.quantize_4bit(W, blocksize=64, quant_type="nf4")
    wp_ref, st_ref = F._convert_weight_packed_for_cpu(q_ref.clone(), st)
    assert torch.equal(wp, wp_ref) and torch.equal(amt, st_ref.absmax)
    for M in (1, 8):
        x = torch.randn(M, K).bfloat16()
        y = O.ref_fused_gemv(x, wp, amt, N, K, 64, "nf4")
        y32 = O.gemm_4bit(x, q, (N, K), am, 64, "nf4")[1]
        assert (y.float() - y32).norm() / y32.norm() < 1e-2


# ------------------------------------------------------------------------------------------ host layer vs live reference
# In this interpreter the reference owns the ``bitsandbytes::`` ops and their CPU kernels, so running
# bitsandbytes_amd's modules on CPU tensors exercises OUR host logic (dtype policy, bias handling, quant-state
# plumbing, state-dict layout) on top of the REFERENCE's arithmetic: results must match the reference's own
# modules bit for bit.
def _pair_linear(ref_bnb, compress_statistics, quant_type, storage, bias=True):
    import bitsandbytes_amd.nn as mnn

    torch.manual_seed(11)
    fp = torch.nn.Linear(192, 64, bias=bias)
    theirs = ref_bnb.nn.Linear4bit(192, 64, bias=bias, compress_statistics=compress_statistics, quant_type=quant_type,
                                   quant_storage=storage)
    mine = mnn.Linear4bit(192, 64, bias=bias, compress_statistics=compress_statistics, quant_type=quant_type,
                          quant_storage=storage)
    theirs.load_state_dict(fp.state_dict())
    mine.load_state_dict(fp.state_dict())
    return theirs.to("cpu"), mine.to("cpu")


@pytest.mark.parametrize("quant_type", ["nf4", "fp4"])
@pytest.mark.parametrize("compress_statistics", [False, True])
@pytest.mark.parametrize("storage", [torch.uint8, torch.bfloat16], ids=["u8", "bf16storage"])
def test_linear4bit_module_matches_reference_module(ref, quant_type, compress_statistics, storage):
    bnb, _ = ref
    theirs, mine = _pair_linear(bnb, compress_statistics, quant_type, storage)
    assert torch.equal(theirs.weight.data.view(torch.uint8), mine.weight.data.view(torch.uint8))
    assert mine.weight.dtype == theirs.weight.dtype and mine.weight.shape == theirs.weight.shape
    for x in (torch.randn(1, 192), torch.randn(3, 5, 192), torch.randn(7, 192).bfloat16(), torch.randn(2, 192).half()):
        y_t, y_m = theirs(x), mine(x)
        assert y_t.dtype == y_m.dtype and y_t.shape == y_m.shape
        assert torch.equal(y_t, y_m)
    # state dicts: same keys, same bytes; each implementation loads the other's
    sd_t, sd_m = theirs.state_dict(), mine.state_dict()
    assert set(sd_t) == set(sd_m)
    for k in sd_t:  # byte comparison: packed weights viewed as bf16 storage contain NaN bit patterns
        assert sd_t[k].dtype == sd_m[k].dtype and sd_t[k].shape == sd_m[k].shape, k
        assert torch.equal(sd_t[k].contiguous().view(torch.uint8), sd_m[k].contiguous().view(torch.uint8)), k
    import bitsandbytes_amd.nn as mnn

    stats = {k[len("weight."):]: v for k, v in sd_t.items() if k.startswith("weight.")}
    w_mine = mnn.Params4bit.from_prequantized(sd_t["weight"], dict(stats), device="cpu")
    w_theirs = bnb.nn.Params4bit.from_prequantized(sd_m["weight"], dict(stats), device="cpu")
    assert torch.equal(w_mine.data.view(torch.uint8), w_theirs.data.view(torch.uint8))
    for f in ("absmax", "code", "blocksize", "quant_type", "dtype", "shape", "nested"):
        a, b = getattr(w_mine.quant_state, f), getattr(w_theirs.quant_state, f)
        assert torch.equal(a, b) if isinstance(a, torch.Tensor) else a == b, f


def test_matmul_4bit_backward_matches_reference(ref):
    bnb, F = ref
    import bitsandbytes_amd as mine

    W = (torch.randn(48, 128) / 11).bfloat16()
    q, st = F.quantize_4bit(W, quant_type="nf4")
    grads = []
    for impl in (bnb, mine):
        x = torch.randn(6, 128, dtype=torch.bfloat16, requires_grad=True)
        torch.manual_seed(0)
        x.data.copy_(torch.randn(6, 128).bfloat16())
        y = impl.matmul_4bit(x, q.t(), st)  # [K, N] orientation, as Linear4bit passes it
        y.float().pow(2).sum().backward()
        grads.append((y.detach().clone(), x.grad.clone()))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][1], grads[1][1])


@pytest.mark.parametrize("cls", ["EmbeddingNF4", "EmbeddingFP4"])
@pytest.mark.parametrize("dim", [64, 72])
def test_embedding4bit_matches_reference_module(ref, cls, dim):
    bnb, _ = ref
    import bitsandbytes_amd.nn as mnn

    # the fused row-gather op is ours; on CPU it is played by the oracle (test infrastructure)
    try:
        @torch.library.register_kernel("bitsandbytes_amd::dequantize_4bit_rows", "cpu")
        def _(A, absmax, indices, row_len, blocksize, quant_type, dtype):
            n_rows = A.numel() * A.element_size() * 2 // row_len
            packed = A.contiguous().view(torch.uint8).view(n_rows, row_len // 2)
            idx = indices.reshape(-1).long()
            return O.dequantize_4bit(packed[idx].reshape(-1, 1), absmax.view(n_rows, -1)[idx].reshape(-1), blocksize,
                                     quant_type, (*indices.shape, row_len), dtype)
    except RuntimeError:
        pass  # already registered by an earlier parametrisation
    torch.manual_seed(5)
    fp = torch.nn.Embedding(40, dim)
    theirs, mine = getattr(bnb.nn, cls)(40, dim), getattr(mnn, cls)(40, dim)
    theirs.load_state_dict(fp.state_dict())
    mine.load_state_dict(fp.state_dict())
    theirs, mine = theirs.to("cpu"), mine.to("cpu")
    assert torch.equal(theirs.weight.data, mine.weight.data)
    idx = torch.tensor([[0, 39, 3], [3, 3, 17]])
    assert same_values(mine(idx), theirs(idx))


@pytest.mark.parametrize("compress_statistics", [False, True])
def test_parametrize_matches_reference(ref, compress_statistics):
    bnb, _ = ref
    import bitsandbytes.nn.parametrize as rp

    import bitsandbytes_amd.nn.parametrize as mp

    class Experts(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(9)
            self.w = torch.nn.Parameter(torch.randn(3, 32, 64) * 0.1)

    a, b = Experts(), Experts()
    rp.replace_parameter_4bit(a, "w", compress_statistics=compress_statistics, quant_type="nf4")
    mp.replace_parameter_4bit(b, "w", compress_statistics=compress_statistics, quant_type="nf4")
    assert torch.equal(a.w, b.w)
    sd_a, sd_b = a.state_dict(), b.state_dict()
    assert set(sd_a) == set(sd_b)
    for k in sd_a:
        assert torch.equal(sd_a[k], sd_b[k]), k


@pytest.mark.parametrize("quant_type", ["nf4", "fp4"])
@pytest.mark.parametrize("compress_statistics", [False, True])
@pytest.mark.parametrize("storage", [torch.uint8, torch.float16, torch.float32], ids=["u8", "f16storage", "f32storage"])
def test_functional_wrappers_match_reference(ref, quant_type, compress_statistics, storage):
    """functional.quantize_4bit / dequantize_4bit / gemv_4bit / (de)quantize_blockwise: same outputs, same
    QuantState contents, same shapes and dtypes as the reference wrappers."""
    _, F = ref
    import bitsandbytes_amd.functional as MF

    A = (torch.randn(96, 160) * 0.3).half()
    q_r, st_r = F.quantize_4bit(A, blocksize=64, compress_statistics=compress_statistics, quant_type=quant_type,
                                quant_storage=storage)
    q_m, st_m = MF.quantize_4bit(A, blocksize=64, compress_statistics=compress_statistics, quant_type=quant_type,
                                 quant_storage=storage)
    assert q_r.dtype == q_m.dtype and q_r.shape == q_m.shape
    assert torch.equal(q_r.contiguous().view(torch.uint8), q_m.contiguous().view(torch.uint8))
    for f in ("absmax", "code", "blocksize", "quant_type", "dtype", "shape", "nested", "offset"):
        a, b = getattr(st_r, f), getattr(st_m, f)
        assert torch.equal(a, b) if isinstance(a, torch.Tensor) else a == b, f
    if compress_statistics:
        for f in ("absmax", "code", "blocksize", "dtype"):
            a, b = getattr(st_r.state2, f), getattr(st_m.state2, f)
            assert torch.equal(a, b) if isinstance(a, torch.Tensor) else a == b, f"state2.{f}"
    d_r, d_m = F.dequantize_4bit(q_r, st_r), MF.dequantize_4bit(q_m, st_m)
    assert d_r.dtype == d_m.dtype and d_r.shape == d_m.shape and same_values(d_r, d_m)
    # absmax= / blocksize= calling convention (no QuantState)
    if not compress_statistics:
        for mod, q, st in ((F, q_r, st_r), (MF, q_m, st_m)):
            with pytest.raises(ValueError):  # both require `out` when no QuantState is given
                mod.dequantize_4bit(q, absmax=st.absmax, blocksize=64, quant_type=quant_type)
        # (the out= form has no CPU kernel in the reference either; the GPU suite covers it on the HIP path)
    # 8-bit pair through the wrappers
    v = torch.randn(1000) * 0.01
    c_r, s_r = F.quantize_blockwise(v, blocksize=256)
    c_m, s_m = MF.quantize_blockwise(v, blocksize=256)
    assert torch.equal(c_r, c_m) and torch.equal(s_r.absmax, s_m.absmax) and torch.equal(s_r.code, s_m.code)
    assert same_values(F.dequantize_blockwise(c_r, s_r), MF.dequantize_blockwise(c_m, s_m))


def test_replace_linear_matches_reference(ref):
    """utils.replace_linear swaps exactly the modules the reference's does (same walk order, skip list, bias flag)."""
    bnb_ref, _ = ref
    from bitsandbytes_amd.utils import replace_linear as mine

    def tree():
        torch.manual_seed(0)
        inner = torch.nn.Sequential(torch.nn.Linear(16, 32, bias=False), torch.nn.ReLU(), torch.nn.Linear(32, 16))
        m = torch.nn.Sequential()
        m.add_module("inner", inner)
        m.add_module("proj", torch.nn.Linear(16, 16))
        m.add_module("lm_head", torch.nn.Linear(16, 4))
        return m

    class Marker(torch.nn.Linear):
        pass

    calls = {"mine": [], "ref": []}
    a = mine(tree(), lambda i, o, b: (calls["mine"].append((i, o, b)), Marker(i, o, b))[1], copy_weights=True)
    b = bnb_ref.utils.replace_linear(tree(), lambda i, o, b: (calls["ref"].append((i, o, b)), Marker(i, o, b))[1],
                                     copy_weights=True)
    assert calls["mine"] == calls["ref"]
    assert [(n, type(m).__name__) for n, m in a.named_modules()] == [(n, type(m).__name__) for n, m in b.named_modules()]
    for (_, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(pa, pb)


def test_mode_a_reference_python_binds_our_library():
    """INTEGRATION.md mode A, as far as it goes without a GPU: the reference's own loader classes wrap OUR shared
    library, classify it as a GPU build (get_context / cget_managed_ptr), its unmodified
    bitsandbytes/backends/cuda/ops.py imports against it (every _setup_ctypes call, :16-66), the 4-bit / 8-bit
    symbols resolve to real functions with the argtypes the reference assigns, symbols outside this path become the
    reference's raise-on-call stubs, and the reference's CUDA-key kernels get registered. Runs in a subprocess: the
    reference's "cuda" kernels must not be registered in the interpreter that also imports bitsandbytes_amd."""
    import subprocess
    import sys
    import textwrap

    from conftest import ROOT

    script = textwrap.dedent(f"""
        import ctypes as ct, importlib, sys
        sys.path.insert(0, {ROOT!r})
        import torch
        from oracle.ref_import import import_reference
        bnb = import_reference()
        ce = bnb.cextension
        dll = ct.cdll.LoadLibrary({ROOT!r} + "/bitsandbytes_amd/libbitsandbytes_mi355x.so")
        assert hasattr(dll, "get_context") and hasattr(dll, "cget_managed_ptr")   # cextension.py:371-372
        wrapped = ce.CudaBNBNativeLibrary(dll)
        assert wrapped.compiled_with_cuda
        ce.lib = wrapped
        bnb.lib = wrapped
        ops = importlib.import_module("bitsandbytes.backends.cuda.ops")
        assert ops.lib is wrapped
        for d in ("fp32", "fp16", "bf16"):
            for name, nargs in ((f"cgemm_4bit_{{d}}", 14), (f"cgemm_4bit_inference_naive_{{d}}", 13),
                                (f"cquantize_blockwise_{{d}}", 6), (f"cdequantize_blockwise_{{d}}", 7),
                                (f"cquantize_blockwise_{{d}}_nf4", 6), (f"cdequantize_blockwise_{{d}}_fp4", 7)):
                fn = getattr(wrapped, name)
                assert isinstance(fn, ct._CFuncPtr), name
                assert len(fn.argtypes) == nargs, (name, len(fn.argtypes))
        stub = wrapped.cigemmlt_32
        assert not isinstance(stub, ct._CFuncPtr)
        try:
            stub()
            raise SystemExit("stub did not raise")
        except RuntimeError:
            pass
        for op in ("gemm_4bit", "gemv_4bit", "quantize_4bit", "dequantize_4bit", "quantize_blockwise", "dequantize_blockwise"):
            assert torch._C._dispatch_has_kernel_for_dispatch_key(f"bitsandbytes::{{op}}", "CUDA"), op
        assert wrapped.get_context() not in (None, 0)
        print("MODE_A_OK")
    """)
    proc = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600)
    assert "MODE_A_OK" in proc.stdout, proc.stdout[-1500:] + proc.stderr[-2500:]
