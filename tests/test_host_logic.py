"""CPU: host-side logic of the package — op schemas/fake kernels, QuantState (de)serialisation,
functional wrappers' validation, Linear4bit / Params4bit behaviour and state-dict round trips.
Arithmetic on CPU tensors is provided by the ORACLE, registered as test-only CPU kernels
(tests/_oracle_cpu_backend.py); the product itself has no CPU kernels."""
import copy
import io
import pickle

import pytest
import torch

import _oracle_cpu_backend
import bitsandbytes_amd as bnb
import bitsandbytes_amd.functional as F
from bitsandbytes_amd.nn import Embedding4bit, EmbeddingFP4, EmbeddingNF4, Linear4bit, LinearFP4, LinearNF4, Params4bit
from bitsandbytes_amd.nn import parametrize as bnb_parametrize
from conftest import from_bits, golden, rel_err

_oracle_cpu_backend.register()


def test_code_tables_match_reference_golden():
    G = golden()
    assert torch.equal(F.get_4bit_type("nf4", device="cpu"), from_bits(G["code/nf4"], 0))
    assert torch.equal(F.get_4bit_type("fp4", device="cpu"), from_bits(G["code/fp4"], 0))
    assert torch.equal(F.create_dynamic_map().view(torch.int32), from_bits(G["code/dynamic"], 0).view(torch.int32))


@pytest.mark.parametrize("storage", [torch.uint8, torch.bfloat16, torch.float32])
def test_fake_kernels_shapes(storage):
    A = torch.empty(48, 128, device="meta", dtype=torch.bfloat16)
    q, am = torch.ops.bitsandbytes.quantize_4bit.default(A, 64, "nf4", storage)
    assert q.shape == ((48 * 128 + 1) // (storage.itemsize * 2), 1) and q.dtype == storage
    assert am.shape == (96,) and am.dtype == torch.float32
    d = torch.ops.bitsandbytes.dequantize_4bit.default(q, am, 64, "nf4", (48, 128), torch.float16)
    assert d.shape == (48, 128) and d.dtype == torch.float16
    x = torch.empty(3, 5, 128, device="meta", dtype=torch.bfloat16)
    y = torch.ops.bitsandbytes.gemm_4bit.default(x, q, (48, 128), am, 64, "nf4")
    assert y.shape == (3, 5, 48)
    with pytest.raises(RuntimeError):
        torch.ops.bitsandbytes.quantize_4bit.default(A, 48, "nf4", storage)
    with pytest.raises(RuntimeError):
        torch.ops.bitsandbytes.gemm_4bit.default(x, q, (48, 128), am, 64, "int4")


def test_nested_quantize_operator_shapes_and_per_device_constants():
    """The one-call nested quantize (bitsandbytes_amd::quantize_4bit_nested): shapes of its four results on the meta device; and the
    constants the functional layer hands out (4-bit code table, dynamic map) are per-device singletons whose copies are independent."""
    A = torch.empty(1000, 96, device="meta", dtype=torch.float16)
    code8 = torch.empty(256, device="meta", dtype=torch.float32)
    q, q_am, am2, off = torch.ops.bitsandbytes_amd.quantize_4bit_nested.default(A, code8, 64, "fp4", torch.uint8)
    assert q.shape == (48000, 1) and q_am.shape == (1500,) and q_am.dtype == torch.uint8
    assert am2.shape == (6,) and am2.dtype == torch.float32 and off.shape == () and off.dtype == torch.float32
    with pytest.raises(RuntimeError):
        torch.ops.bitsandbytes_amd.quantize_4bit_nested.default(A, code8[:16], 64, "fp4", torch.uint8)
    d = torch.ops.bitsandbytes_amd.dequantize_4bit_nested.default(q, q_am, am2, code8, off, 64, "fp4", [1000, 96], torch.float16)
    assert d.shape == (1000, 96) and d.dtype == torch.float16
    with pytest.raises(RuntimeError):
        torch.ops.bitsandbytes_amd.dequantize_4bit_nested.default(q, am2, am2, code8, off, 64, "fp4", [1000, 96], torch.float16)
    a = F.get_4bit_type("nf4", device="cpu")
    a.mul_(2.0)  # a caller's copy: the next call must not see it
    b = F.get_4bit_type("nf4", device="cpu")
    assert b.abs().max().item() == 1.0 and b.data_ptr() != a.data_ptr()
    assert torch.equal(b, from_bits(golden()["code/nf4"], 0))
    d1, d2 = F._dynamic_map("cpu"), F._dynamic_map(torch.device("cpu"))
    assert d1.data_ptr() == d2.data_ptr() and torch.equal(d1, F.create_dynamic_map())
    # states built from it carry copies
    _, st = F.quantize_blockwise(torch.randn(512), blocksize=256)
    assert st.code.data_ptr() != d1.data_ptr() and torch.equal(st.code, d1)


def test_functional_validation_errors():
    A = torch.randn(64, 64)
    with pytest.raises(ValueError, match="invalid blocksize"):
        F.quantize_4bit(A, blocksize=48)
    with pytest.raises(ValueError, match="quant_type"):
        F.quantize_4bit(A, quant_type="int4")
    with pytest.raises(ValueError, match="16/32-bit floats"):
        F.quantize_4bit(A.to(torch.int32))
    with pytest.raises(ValueError, match="requires both absmax and out"):
        F.dequantize_4bit(torch.zeros(8, 1, dtype=torch.uint8))
    with pytest.raises(ValueError, match="state cannot be None"):
        F.gemv_4bit(A[:1], torch.zeros(8, 1, dtype=torch.uint8))
    with pytest.raises(ValueError, match="quant_state is required"):
        bnb.matmul_4bit(A, torch.zeros(8, 1, dtype=torch.uint8), None)


@pytest.mark.parametrize("quant_type", ["nf4", "fp4"])
@pytest.mark.parametrize("double_quant", [False, True])
def test_quant_state_roundtrip(quant_type, double_quant):
    W = torch.randn(64, 256).bfloat16()
    q, st = F.quantize_4bit(W, blocksize=64, quant_type=quant_type, compress_statistics=double_quant)
    assert st.nested == double_quant and st.shape == W.shape and st.dtype == torch.bfloat16
    d = st.as_dict(packed=True)
    assert f"quant_state.bitsandbytes__{quant_type}" in d and all(isinstance(v, torch.Tensor) for v in d.values())
    st2 = F.QuantState.from_dict(dict(d), device="cpu")
    assert st2 == st
    assert torch.equal(F.dequantize_4bit(q, st2), F.dequantize_4bit(q, st))
    # legacy list view
    assert st[0] is st.absmax and st[3] == 64 and st[5] == quant_type
    # error envelope of the reference's test_4bit_quant (tests/test_functional.py:606-651), blocksize 64
    err = (F.dequantize_4bit(q, st).float() - W.float()).abs().mean().item()
    assert err < (0.0735 if quant_type == "nf4" else 0.098) * 1.15 * (1.1 if double_quant else 1.0)


def test_out_and_legacy_keyword_arguments():
    """out= buffers are written in place and returned; the pre-QuantState keyword form (absmax=, code=, blocksize=)
    gives the same values; nested 8-bit states survive as_dict / from_dict (reference functional.py:613-769,
    :992-1077, :1300-1334; tests/test_functional.py:113-172)."""
    A = torch.randn(4096)
    out = torch.empty(4096, dtype=torch.uint8)
    C, S = F.quantize_blockwise(A, out=out)
    assert C.data_ptr() == out.data_ptr()
    buf = torch.empty(4096)
    D = F.dequantize_blockwise(C, S, out=buf)
    assert D.data_ptr() == buf.data_ptr()
    assert torch.equal(D, F.dequantize_blockwise(C, absmax=S.absmax, code=S.code, blocksize=4096))
    for dtype in (torch.float32, torch.float16, torch.bfloat16):
        X = torch.randn(256, 256).to(dtype)
        Cn, Sn = F.quantize_blockwise(X, blocksize=256, nested=True)
        assert Sn.nested and Sn.state2 is not None and Sn.offset is not None
        Sr = F.QuantState.from_dict(Sn.as_dict(), device=torch.device("cpu"))
        Y = F.dequantize_blockwise(Cn, Sr)
        assert Y.dtype == dtype and (X.float() - Y.float()).abs().mean() < 0.011
    W = torch.randn(64, 128).bfloat16()
    q, st = F.quantize_4bit(W, quant_type="nf4")
    o = torch.empty(64, 128, dtype=torch.bfloat16)
    d = F.dequantize_4bit(q, st, out=o)
    assert d.data_ptr() == o.data_ptr() and torch.equal(d, F.dequantize_4bit(q, st))
    o2 = torch.empty(64, 128, dtype=torch.bfloat16)
    assert torch.equal(d, F.dequantize_4bit(q, absmax=st.absmax, out=o2, blocksize=64, quant_type="nf4"))
    with pytest.raises(ValueError, match="both absmax and out"):
        F.dequantize_4bit(q, absmax=st.absmax, blocksize=64, quant_type="nf4")
    x = torch.randn(1, 128).bfloat16()
    y = torch.empty(1, 64, dtype=torch.bfloat16)
    r = F.gemv_4bit(x, q.t(), out=y, state=st)
    assert r.data_ptr() == y.data_ptr() and torch.equal(r, F.gemv_4bit(x, q.t(), state=st))


def test_quant_storage_views_are_byte_identical():
    W = torch.randn(32, 64).half()
    q8, _ = F.quantize_4bit(W, quant_type="nf4", quant_storage=torch.uint8)
    for st_dtype in (torch.bfloat16, torch.float16, torch.float32):
        q, state = F.quantize_4bit(W, quant_type="nf4", quant_storage=st_dtype)
        assert q.dtype == st_dtype and q.shape == (32 * 64 // (2 * st_dtype.itemsize), 1)
        assert torch.equal(q.view(torch.uint8).reshape(-1), q8.reshape(-1))
        assert torch.equal(F.dequantize_4bit(q, state), F.dequantize_4bit(q8, state))


def _make_layer(cls=Linear4bit, bias=True, **kw):
    torch.manual_seed(1)
    ref = torch.nn.Linear(128, 48, bias=bias)
    layer = cls(128, 48, bias=bias, **kw)
    layer.load_state_dict(ref.state_dict())
    return ref, layer.to("cpu")  # first .to(device) quantises (lazy quantisation contract)


@pytest.mark.parametrize("cls,kw", [(LinearNF4, {}), (LinearFP4, {}), (Linear4bit, {"quant_type": "nf4", "compress_statistics": False})])
def test_linear4bit_forward_close_to_fp(cls, kw):
    ref, layer = _make_layer(cls, **kw)
    assert layer.weight.bnb_quantized and layer.weight.dtype == torch.uint8 and layer.weight.shape == (128 * 48 // 2, 1)
    x = torch.randn(5, 128)
    y = layer(x)
    assert y.shape == (5, 48) and y.dtype == x.dtype
    assert rel_err(y, ref(x)) < 0.2
    # weight orientation: B and B.t() must give identical results (reference test_functional.py:1016-1034)
    st = layer.weight.quant_state
    assert torch.equal(bnb.matmul_4bit(x, layer.weight.data, st), bnb.matmul_4bit(x, layer.weight.data.t(), st))


@pytest.mark.parametrize("double_quant", [False, True])
@pytest.mark.parametrize("storage", [torch.uint8, torch.bfloat16])
def test_linear4bit_state_dict_roundtrip(double_quant, storage):
    _, layer = _make_layer(Linear4bit, quant_type="nf4", compress_statistics=double_quant, quant_storage=storage)
    sd = layer.state_dict()
    keys = set(sd)
    assert {"weight", "bias", "weight.absmax", "weight.quant_map", "weight.quant_state.bitsandbytes__nf4"} <= keys
    assert ("weight.nested_absmax" in keys) == double_quant
    buf = io.BytesIO()
    torch.save(sd, buf)
    buf.seek(0)
    sd2 = torch.load(buf)
    # rebuild the way HF loaders do: Params4bit.from_prequantized(weight, {the other weight.* items})
    new = Linear4bit(128, 48, quant_type="nf4", compress_statistics=double_quant, quant_storage=storage)
    stats = {k[len("weight."):]: v for k, v in sd2.items() if k.startswith("weight.")}
    new.weight = Params4bit.from_prequantized(sd2["weight"], stats, device="cpu", module=new)
    new.bias.data = sd2["bias"]
    x = torch.randn(3, 128)
    assert torch.equal(new(x), layer(x))
    assert new.weight.quant_state == layer.weight.quant_state


def test_params4bit_copy_pickle_chunk():
    _, layer = _make_layer(Linear4bit, quant_type="nf4")
    w = layer.weight
    for clone in (copy.copy(w), copy.deepcopy(w), pickle.loads(pickle.dumps(w))):
        assert isinstance(clone, Params4bit) and clone.quant_type == "nf4" and clone.bnb_quantized
        assert torch.equal(clone.data, w.data) and clone.quant_state == w.quant_state
    parts = torch.chunk(w, 2, dim=0)
    assert all(isinstance(p, Params4bit) and p.quant_state is w.quant_state and p.blocksize == w.blocksize for p in parts)
    assert torch.equal(torch.cat([p.data for p in parts]), w.data)
    # FSDP-style attribute proxies
    assert w.absmax is w.quant_state.absmax and w.quant_map is w.quant_state.code
    assert w.nested_absmax is w.quant_state.state2.absmax


def test_quant_state_recovered_after_param_replacement():
    """FSDP replaces the parameter by a plain tensor; forward must restore quant_state from the module."""
    _, layer = _make_layer(Linear4bit, quant_type="nf4")
    x = torch.randn(2, 128)
    y0 = layer(x)
    layer.weight = torch.nn.Parameter(layer.weight.data.clone(), requires_grad=False)
    assert torch.equal(layer(x), y0) and isinstance(layer.weight, Params4bit)


def test_matmul_4bit_backward_matches_dequant_linear():
    _, layer = _make_layer(Linear4bit, quant_type="nf4", compress_statistics=False, bias=True)
    st = layer.weight.quant_state
    x = torch.randn(4, 128, requires_grad=True)
    bias = layer.bias.detach().clone().requires_grad_(True)
    y = bnb.matmul_4bit(x, layer.weight, st, bias=bias)
    y.sum().backward()
    W = F.dequantize_4bit(layer.weight.data, st).float()
    assert torch.allclose(x.grad, torch.ones(4, 48) @ W, atol=1e-5)
    assert torch.allclose(bias.grad, torch.full((48,), 4.0))


def test_legacy_kn_orientation_warns_and_works():
    W = torch.randn(128, 48)  # [K, N]
    q, st = F.quantize_4bit(W, quant_type="nf4")
    x = torch.randn(2, 128)
    with pytest.warns(DeprecationWarning):
        y = bnb.matmul_4bit(x, q, st)
    assert y.shape == (2, 48)


def test_empty_input():
    _, layer = _make_layer(Linear4bit, quant_type="nf4")
    y = bnb.matmul_4bit(torch.empty(0, 128), layer.weight, layer.weight.quant_state)
    assert y.shape == (0, 48)


# ------------------------------------------------------------------------------------------ Embedding4bit / parametrize
@pytest.mark.parametrize("cls,qt", [(EmbeddingNF4, "nf4"), (EmbeddingFP4, "fp4")])
@pytest.mark.parametrize("dim", [64, 192, 72])  # 72: not a multiple of the blocksize -> whole-table path
def test_embedding4bit_matches_dequantized_table(cls, qt, dim):
    """reference tests/test_modules.py (embedding lookups): the lookup equals indexing the dequantized table."""
    torch.manual_seed(3)
    fp = torch.nn.Embedding(50, dim)
    emb = cls(50, dim)
    emb.load_state_dict(fp.state_dict())
    emb = emb.to("cpu")
    assert emb.weight.bnb_quantized and emb.weight.quant_state.quant_type == qt and not emb.weight.quant_state.nested
    idx = torch.tensor([[0, 49, 7], [7, 7, 23]])
    out = emb(idx)
    table = F.dequantize_4bit(emb.weight.data, emb.weight.quant_state)
    assert out.shape == (2, 3, dim) and out.dtype == fp.weight.dtype
    assert torch.equal(out, table[idx])
    assert rel_err(out, fp(idx)) < 0.25
    assert torch.equal(emb(idx.int()), out)
    with pytest.raises(NotImplementedError):
        emb.state_dict()


def test_embedding4bit_recovers_quant_state_from_module():
    emb = EmbeddingNF4(16, 64).to("cpu")
    idx = torch.arange(16)
    want = emb(idx)
    emb.quant_state = emb.weight.quant_state
    emb.weight = torch.nn.Parameter(emb.weight.data.clone(), requires_grad=False)  # what FSDP-style wrappers do
    assert torch.equal(emb(idx), want)


@pytest.mark.parametrize("quant_type", ["nf4", "fp4"])
@pytest.mark.parametrize("compress_statistics", [False, True])
def test_replace_parameter_4bit_roundtrip(quant_type, compress_statistics):
    """reference tests/test_parametrize.py: attribute reads give the dequantized tensor; the state dict has
    the Linear4bit key layout and reloads through replace_parameter_4bit_prequantized."""

    class Experts(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.randn(4, 32, 64) * 0.1)
            self.reads = 0

        def forward(self, x):
            a = self.w  # two reads inside one forward: the second one must come from the cache
            b = self.w
            assert a is b
            return torch.einsum("bi,eoi->beo", x, a)

    torch.manual_seed(5)
    m = Experts()
    w_fp = m.w.detach().clone()
    q_want, st_want = F.quantize_4bit(w_fp, quant_type=quant_type, compress_statistics=compress_statistics)
    bnb_parametrize.replace_parameter_4bit(m, "w", compress_statistics=compress_statistics, quant_type=quant_type)
    assert m.w.shape == w_fp.shape and m.w.dtype == w_fp.dtype
    assert torch.equal(m.w, F.dequantize_4bit(q_want, st_want))
    assert rel_err(m.w, w_fp) < 0.2
    assert m.parametrizations.w.original.dtype == torch.uint8
    y = m(torch.randn(3, 64))
    assert y.shape == (3, 4, 32)
    import torch.nn.utils.parametrize as P

    assert P._cache_enabled == 0 and not P._cache  # cache released after the forward

    sd = m.state_dict()
    assert "w" in sd and sd["w"].dtype == torch.uint8 and "parametrizations.w.original" not in sd
    assert f"w.quant_state.bitsandbytes__{quant_type}" in sd and "w.absmax" in sd
    assert ("w.nested_absmax" in sd) == compress_statistics

    # reload into a fresh module holding the packed bytes
    m2 = Experts()
    m2.w = torch.nn.Parameter(sd["w"].clone(), requires_grad=False)
    qs = {k[len("w."):]: v for k, v in sd.items() if k.startswith("w.")}
    bnb_parametrize.replace_parameter_4bit_prequantized(m2, "w", qs, device=torch.device("cpu"))
    assert torch.equal(m2.w, m.w)

    with pytest.raises(AttributeError):
        bnb_parametrize.replace_parameter_4bit(m, "nope")
    m.buf = torch.zeros(3)
    with pytest.raises(TypeError):
        bnb_parametrize.replace_parameter_4bit(m, "buf")


def test_parametrize_cache_counter_survives_failing_forward():
    import torch.nn.utils.parametrize as P

    class Boom(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.randn(64, 64))

        def forward(self, x):
            _ = self.w
            raise RuntimeError("boom")

    m = Boom()
    bnb_parametrize.replace_parameter_4bit(m, "w")
    for _ in range(3):
        with pytest.raises(RuntimeError):
            m(torch.zeros(1))
    assert P._cache_enabled == 0 and not P._cache


def test_linear4bit_traces_under_torch_compile():
    """The fake kernels are enough for dynamo/AOT to trace Linear4bit with no graph break (the arithmetic in
    this CPU test is the oracle's; the GPU suite repeats it on the HIP kernels)."""
    torch.manual_seed(4)
    torch._dynamo.reset()
    net = torch.nn.Sequential(LinearNF4(128, 64, compute_dtype=torch.bfloat16), torch.nn.GELU(),
                              LinearNF4(64, 128, compute_dtype=torch.bfloat16)).to("cpu")
    x = torch.randn(3, 128, dtype=torch.bfloat16)
    want = net(x)
    compiled = torch.compile(net, backend="aot_eager", fullgraph=True)
    with torch.no_grad():
        assert torch.equal(compiled(x), want)


def test_replace_linear_walks_the_module_tree():
    """utils.replace_linear: nested children, skip list, bias flag, copy_weights (reference utils.py:121-163)."""
    from bitsandbytes_amd.utils import pack_dict_to_tensor, replace_linear, unpack_tensor_to_dict

    class Block(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1 = torch.nn.Linear(64, 128, bias=False)
            self.act = torch.nn.GELU()
            self.fc2 = torch.nn.Linear(128, 64)

    model = torch.nn.Sequential()
    model.add_module("body", torch.nn.ModuleList([Block(), Block()]))
    model.add_module("lm_head", torch.nn.Linear(64, 10))
    out = replace_linear(model, lambda i, o, b: Linear4bit(i, o, b, quant_type="nf4", compute_dtype=torch.float32))
    assert out is model and type(model.lm_head) is torch.nn.Linear
    for blk in model.body:
        assert isinstance(blk.fc1, Linear4bit) and blk.fc1.bias is None and (blk.fc1.in_features, blk.fc1.out_features) == (64, 128)
        assert isinstance(blk.fc2, Linear4bit) and blk.fc2.bias is not None and isinstance(blk.act, torch.nn.GELU)
    plain = torch.nn.Sequential(torch.nn.Linear(8, 8))
    w = plain[0].weight
    replace_linear(plain, lambda i, o, b: torch.nn.Linear(i, o, b), skip_modules=(), copy_weights=True)
    assert plain[0].weight is w
    d = {"quant_type": "nf4", "blocksize": 64, "shape": [4, 8], "nested": {"a": 1.5}}
    assert unpack_tensor_to_dict(pack_dict_to_tensor(d)) == d


@pytest.mark.parametrize("double_quant", [False, True])
def test_state_dict_travels_through_safetensors(tmp_path, double_quant):
    """The packed state-dict form holds tensors only (the QuantState's non-tensor fields ride in a uint8 JSON blob,
    reference functional.py:545-578), so a quantised layer round-trips through a .safetensors file and
    Params4bit.from_prequantized rebuilds it - the way HF NF4 checkpoints are stored."""
    st = pytest.importorskip("safetensors.torch")
    torch.manual_seed(0)
    layer = Linear4bit(128, 64, bias=True, quant_type="nf4", compress_statistics=double_quant, compute_dtype=torch.bfloat16)
    layer = layer.to("cpu")  # quantises on the move (test-only oracle backend), like .to("cuda") does
    assert layer.weight.quant_state is not None and layer.weight.dtype == torch.uint8
    sd = {k: v.contiguous() for k, v in layer.state_dict().items()}
    assert all(isinstance(v, torch.Tensor) for v in sd.values())
    path = str(tmp_path / "layer.safetensors")
    st.save_file(sd, path)
    loaded = st.load_file(path)
    assert set(loaded) == set(sd) and all(torch.equal(loaded[k], sd[k]) for k in sd)
    qs = {k[len("weight."):]: v for k, v in loaded.items() if k.startswith("weight.")}
    w = Params4bit.from_prequantized(loaded["weight"], qs, device="cpu")
    assert torch.equal(w.data, layer.weight.data)
    x = torch.randn(3, 128).bfloat16()
    y_ref = layer(x)
    y = bnb.matmul_4bit(x, w, bias=loaded["bias"].to(x.dtype), quant_state=w.quant_state)
    assert torch.equal(y, y_ref)


def test_rt_mfma_lane_algebra_emulation():
    """The index algebra of csrc/gemm4_mfma_rt.hip (coalesced "4r + p" loads, LDS transposition slots, permlane32_swap
    regrouping, activation k order) replayed lane by lane against the hardware semantics of the MFMA - see
    tests/checks/emulate_rt_mfma.py."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location(
        "emulate_rt_mfma", os.path.join(os.path.dirname(__file__), "checks", "emulate_rt_mfma.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for M in (1, 7, 16):
        assert mod.emulate(M=M, K=512, seed=M) < 1e-12
        assert mod.emulate_bs32(M=M, K=512, seed=M) < 1e-12  # blocksize 32: the 4 x 4 lane-group transposition (round 5)
    assert mod.final_sum_mapping_ok()



def test_round4_index_algebra_emulation():
    """Index algebra added in round 4, replayed on the CPU (tests/checks/emulate_round4_index_algebra.py): (1) the K-quarter
    kernel's accumulator-layout slabs against gemm4_finalize_kq_kernel's decoding of them - every element of a workgroup tile
    stored once and attributed to the row / column its lane held; (2) the peer chain's transport - producer granule slot ->
    consumer fetch -> swizzled activation image -> the element a decoding lane reads: every k of x reaches the lane that
    multiplies weight k, for every chain shape the bench and the tests use."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location(
        "emulate_round4_index_algebra", os.path.join(os.path.dirname(__file__), "checks", "emulate_round4_index_algebra.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.check_kq_slabs()
    for (K, world, ns) in ((4096, 1, 4096), (8192, 2, 4096), (8192, 4, 2048), (11008, 8, 1376), (16384, 4, 4096), (16384, 8, 2048), (4096, 8, 512)):
        assert mod.check_peer_chain(K, world, ns)


def test_grad_input_lane_algebra_emulation():
    """The index algebra of csrc/gemm4_grad_input.hip (one dword per weight row and lane, nibble j = B operand of the strided
    column tile {8 c + j}, the swizzled private grad_out patch, the scale patch, the output mapping, the dealing of the final
    sum) replayed lane by lane against the hardware semantics of the MFMA - see tests/checks/emulate_grad_input.py."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location(
        "emulate_grad_input", os.path.join(os.path.dirname(__file__), "checks", "emulate_grad_input.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for (M, N, MT) in ((37, 96, 4), (20, 64, 2), (9, 32, 1)):
        assert mod.emulate(M=M, N=N, MT=MT, seed=M) < 1e-12
    assert mod.patch_swizzle_conflict_free()
    assert all(mod.final_sum_units_cover_the_tile(mt) for mt in (1, 2, 4))


def test_blockwise8_threshold_finders_and_byte_table_emulation():
    """csrc/blockwise8.hip restated in numpy fp32: the inverse-and-walk threshold finder equals the 17-step bisection it
    replaced, and the scatter + prefix-sum byte table equals both the direct count and the reference's own table rule
    (csrc/cpu_ops.cpp:501-520) - for every code map constructor of the package and the edge values of the comparison."""
    import importlib.util
    import os

    import numpy as np

    import bitsandbytes_amd.functional as F

    spec = importlib.util.spec_from_file_location(
        "emulate_q8_thresholds", os.path.join(os.path.dirname(__file__), "checks", "emulate_q8_thresholds.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for code in (F.create_dynamic_map(), F.create_dynamic_map(signed=False), F.create_linear_map(True, 8), F.create_linear_map(True, 2),
                 F.create_fp8_map(True, 4, 3, 8), F.create_fp8_map(True, 2, 1, 4), F.create_normal_map(), F.create_dynamic_map(True, 3, 3)):
        assert mod.check_code(code.numpy())
    for m in (-2.0, -1.0, 0.0, 1.0, 2.0, float("inf"), float("nan")):
        assert mod.first_bin_above(m) == mod.bisect_threshold(np.float32(m))


def test_reference_plugin_point_loads_this_backend():
    """pyproject.toml declares the entry point the reference discovers in `_import_backends()` (bitsandbytes/__init__.py:52-70,
    group "bitsandbytes.backends"). A dist-info directory generated from that stanza is put on sys.path (what an installed wheel
    provides) and the reference's own loader code is run over it: it must find the entry, load it and call it, after which this
    package is imported and its kernels are registered. Where the reference checkout exists, its package is imported for real -
    `import bitsandbytes` runs `_import_backends()` at import time."""
    import os
    import subprocess
    import sys
    import tempfile
    import textwrap

    try:
        import tomllib
    except ModuleNotFoundError:  # Python 3.10
        import tomli as tomllib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "pyproject.toml"), "rb") as fh:
        meta = tomllib.load(fh)
    eps = meta["project"]["entry-points"]["bitsandbytes.backends"]
    assert eps == {"mi355x": "bitsandbytes_amd.backends.plugin:register"}
    site = tempfile.mkdtemp(prefix="bnb_plugin_site_")
    di = os.path.join(site, "bitsandbytes_amd_mi355x-0.1.0.dist-info")
    os.makedirs(di)
    with open(os.path.join(di, "METADATA"), "w") as fh:
        fh.write("Metadata-Version: 2.1\nName: bitsandbytes-amd-mi355x\nVersion: 0.1.0\n")
    with open(os.path.join(di, "entry_points.txt"), "w") as fh:
        fh.write("[bitsandbytes.backends]\n" + "".join(f"{k} = {v}\n" for k, v in eps.items()))
    ref = os.environ.get("BNB_REFERENCE_DIR", "/root/reference")
    have_ref = os.path.isdir(os.path.join(ref, "bitsandbytes"))
    script = textwrap.dedent(f"""
        import sys
        sys.dont_write_bytecode = True
        sys.path[:0] = [{site!r}, {root!r}] + ([{ref!r}] if {have_ref} else [])
        if {have_ref}:
            import bitsandbytes as ref                      # runs _import_backends() at import (reference __init__.py:70)
            assert ref.__file__.startswith({ref!r}), ref.__file__
        else:
            # the loader's own statements (reference bitsandbytes/__init__.py:58-67)
            from importlib.metadata import entry_points
            for ext in entry_points(group="bitsandbytes.backends"):
                ext.load()()
        assert "bitsandbytes_amd" in sys.modules, "the plug-in was not loaded"
        from bitsandbytes_amd.backends import plugin
        assert plugin.REGISTERED
        import torch
        # the HIP ("cuda" key) kernels of the path are this package's
        for op in ("quantize_4bit", "dequantize_4bit", "gemm_4bit", "gemv_4bit", "quantize_blockwise", "dequantize_blockwise"):
            assert torch._C._dispatch_has_kernel_for_dispatch_key("bitsandbytes::" + op, "CUDA"), op
        print("PLUGIN_OK", {have_ref})
    """)
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "PLUGIN_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


class _FakeChain:
    """Stands in for peer.PeerChain on the CPU: records launches, can be told to fail at a given layer."""

    def __init__(self, fail_at=None, refuse_at=None):
        self.world, self.device = 1, torch.device("cpu")
        self.fail_at, self.refuse_at = fail_at, refuse_at
        self.launched, self._broken, self.reads = [], None, 0

    def serves(self, ns, K, blocksize, consume, produce=True):
        return True

    def gemv(self, x, packed, st, bias=None, consume=False, produce=True, dtype=None):
        i = len(self.launched)
        if self.fail_at == i:
            raise ValueError("boom")
        if self.refuse_at == i:
            return False
        self.launched.append((x is None, consume))
        return True

    def read(self, n, dtype):
        self.reads += 1
        return torch.zeros(n, dtype=dtype)


def _chain_shards(bias=False):
    from bitsandbytes_amd.parallel import ShardedLinear4bit

    torch.manual_seed(0)
    shards = []
    for _ in range(3):
        layer = Linear4bit(64, 64, bias=bias, quant_type="nf4", compute_dtype=torch.bfloat16).to("cpu")
        st = layer.weight.quant_state
        shards.append(ShardedLinear4bit(layer.weight.data.view(-1, 1), st, 64, layer.bias))
    return shards


def test_sharded_chain_fused_form_is_refused_when_the_call_wants_gradients():
    """the fused launches record no autograd graph; the member-by-member path (matmul_4bit) does. A call with grad
    enabled and an input (or bias) that requires grad must take the latter - decided in fused(), before any launch."""
    from bitsandbytes_amd.parallel import ShardedLinear4bitChain

    chain = _FakeChain()
    mod = ShardedLinear4bitChain(_chain_shards(), chain)
    x = torch.randn(1, 64).bfloat16()
    with torch.no_grad():
        assert mod.fused(x)
    assert mod.fused(x)  # grad mode on, but nothing requires grad
    assert not mod.fused(x.clone().requires_grad_())
    with torch.no_grad():
        assert mod.fused(x.clone().requires_grad_())  # nothing will be recorded anyway
    assert not mod.fused(torch.randn(2, 64).bfloat16()) and not mod.fused(torch.randn(1, 64))  # two rows / fp32: outside the form
    with_bias = ShardedLinear4bitChain(_chain_shards(bias=True), _FakeChain())
    for s in with_bias.shards:
        s.bias.requires_grad_(False)
    assert with_bias.fused(x)
    with_bias.shards[1].bias.requires_grad_(True)
    assert not with_bias.fused(x)
    # and the gradients really arrive through the member-by-member path
    xg = x.clone().requires_grad_()
    y = mod(xg)
    y.float().sum().backward()
    assert xg.grad is not None and chain.launched == [] and chain.reads == 0


def test_sharded_chain_failure_mid_chain_poisons_the_chain_object():
    """A launch that fails after earlier layers of the same chain went out leaves this rank's exchange count out of step with its
    peers: the chain object must refuse further use loudly. A failure before the first launch leaves it usable."""
    from bitsandbytes_amd.parallel import ShardedLinear4bitChain

    x = torch.randn(1, 64).bfloat16()
    with torch.no_grad():
        first = _FakeChain(fail_at=0)
        with pytest.raises(ValueError):
            ShardedLinear4bitChain(_chain_shards(), first)(x)
        assert first._broken is None
        for chain in (_FakeChain(fail_at=1), _FakeChain(refuse_at=2)):
            with pytest.raises((ValueError, RuntimeError)):
                ShardedLinear4bitChain(_chain_shards(), chain)(x)
            assert chain._broken and "stopped after" in chain._broken and chain.reads == 0
        good = _FakeChain()
        y = ShardedLinear4bitChain(_chain_shards(), good)(x)
        assert y.shape == (1, 64) and good.launched == [(False, False), (True, True), (True, True)] and good.reads == 1


def test_peer_chain_counts_in_fours_and_poisoned_chain_raises():
    """max_values is rounded up to a multiple of 4 (with max_values % 4 == 2 the regions were only 8-byte aligned and
    the kernel's 16-byte quad stores / fetches misaligned) - checked on the class without a GPU; a poisoned chain raises at once."""
    from bitsandbytes_amd.peer import PeerChain

    for given, want in ((2, 4), (4, 4), (30, 32), (32766, 32768), (32768, 32768), (11010, 11012)):
        assert (int(given) + 3) & ~3 == want
    import inspect

    src = inspect.getsource(PeerChain.__init__)
    assert "(int(max_values) + 3) & ~3" in src and "self_test" in src
    chain = PeerChain.__new__(PeerChain)
    chain._broken = "test"
    with pytest.raises(RuntimeError, match="out of step"):
        chain.gemv(None, None, None, consume=True, dtype=torch.bfloat16)
    chain._local = None  # (__del__ of the half-built object)


@pytest.mark.parametrize("double_quant", [False, True], ids=["plain", "nested"])
def test_sharded_ffn_block_host_logic(double_quant):
    """ShardedFFN4bit off the fused form (CPU, a group of one, the oracle's arithmetic): the row-interleaved [gate; up] matrix it would
    hand to the peer chain computes exactly what the two members compute (nested statistics carried un-nested: the same fp32 scales),
    and the block equals down(silu(gate(x)) * up(x)) of the unsharded layers. The fused form's routing rules - one row, 16-bit,
    no gradients wanted, shapes the chain serves - with a duck-typed chain; a failure after the first launch poisons the chain."""
    from bitsandbytes_amd.parallel import ShardedFFN4bit, ShardedLinear4bit

    torch.manual_seed(1)
    H, Fd = 128, 256
    layers = [Linear4bit(k, n, bias=b, quant_type="nf4", compress_statistics=double_quant, compute_dtype=torch.bfloat16).to("cpu")
              for k, n, b in ((H, Fd, True), (H, Fd, False), (Fd, H, True))]
    shards = [ShardedLinear4bit(layer.weight.data.view(-1, 1), layer.weight.quant_state, layer.out_features,
                                None if layer.bias is None else layer.bias.data) for layer in layers]
    ffn = ShardedFFN4bit(*shards)
    gate, up, down = layers
    for M in (1, 3):
        x = torch.randn(M, H).bfloat16()
        with torch.no_grad():
            want = down(torch.nn.functional.silu(gate(x)) * up(x))
            assert torch.equal(ffn(x), want)
            stacked = bnb.matmul_4bit(x, ffn.gu_weight, ffn.gu_state, bias=ffn._stacked_bias(torch.bfloat16))
            assert torch.equal(stacked, torch.stack([gate(x), up(x)], dim=-1).reshape(M, -1))  # rows interleaved: g0, u0, g1, u1, ...
    assert not ffn.gu_state.nested and ffn.gu_state.absmax.dtype == torch.float32 and tuple(ffn.gu_state.shape) == (2 * Fd, H)

    class Chain(_FakeChain):
        def __init__(self, **kw):
            super().__init__(**kw)
            self.gated = []

        def serves(self, ns, K, blocksize, consume, produce=True, gated=False):
            return True

        def gemv(self, x, packed, st, bias=None, consume=False, produce=True, dtype=None, gated=False):
            ok = super().gemv(x, packed, st, bias, consume, produce, dtype)
            if ok:
                self.gated.append(gated)
            return ok

    x1 = torch.randn(1, H).bfloat16()
    with torch.no_grad():
        chain = Chain()
        fused = ShardedFFN4bit(*shards, chain=chain)
        assert fused.fused(x1) and not fused.fused(torch.randn(2, H).bfloat16()) and not fused.fused(torch.randn(1, H))
        y = fused(x1)
        assert y.shape == (1, H) and chain.launched == [(False, False), (True, True)] and chain.gated == [True, False] and chain.reads == 1
        broken = Chain(fail_at=1)
        with pytest.raises(ValueError):
            ShardedFFN4bit(*shards, chain=broken)(x1)
        assert broken._broken and "FFN block stopped after 1" in broken._broken
        first = Chain(fail_at=0)
        with pytest.raises(ValueError):
            ShardedFFN4bit(*shards, chain=first)(x1)
        assert first._broken is None
    assert not ShardedFFN4bit(*shards, chain=Chain()).fused(x1.clone().requires_grad_())
