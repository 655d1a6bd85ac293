"""INTEGRATION.md mode A end to end: the REFERENCE'S OWN host code - bitsandbytes/functional.py, backends/cuda/ops.py,
autograd/_functions.py, nn/modules.py, byte-compiled unmodified by oracle/build_ref.sh - loads this repository's shared
library through its own loader (bitsandbytes/cextension.py:36-57,348-377) and drives real launches on the MI355X:
quantize_4bit -> dequantize_4bit -> gemm_4bit / gemv_4bit / Linear4bit. Results are compared with the CPU oracle.

The compiled reference modules live in oracle/_ref/ref_py (git-ignored build output that travels to the GPU box; the
reference checkout itself does not exist there). Runs in a subprocess: the reference registers kernels for the
``bitsandbytes::`` ops and must not share an interpreter with bitsandbytes_amd.
"""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
from oracle.ref_mode_a import compiled_reference_available  # noqa: E402

PRELUDE = textwrap.dedent(f"""
    import os, sys
    sys.dont_write_bytecode = True
    ROOT = {ROOT!r}
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle.ref_mode_a import make_package
""")


def _run(script, env=None, timeout=600):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, "-c", PRELUDE + textwrap.dedent(script)], capture_output=True, text=True, timeout=timeout,
                       env=e)
    assert r.returncode == 0, f"stdout:\n{r.stdout[-3000:]}\nstderr:\n{r.stderr[-6000:]}"
    return r.stdout


@pytest.mark.skipif(not compiled_reference_available(), reason="oracle/_ref/ref_py not built (oracle/build_ref.sh)")
def test_compiled_reference_package_imports_sourceless():
    """CPU: the byte-compiled reference package imports without its sources and quantizes through the reference's own CPU
    library (what the GPU test relies on, minus the GPU)."""
    out = _run("""
        farm = make_package(os.path.join(ROOT, "oracle", "_ref", "libbitsandbytes_cpu.so"), "libbitsandbytes_cpu.so")
        sys.path.insert(0, farm)
        import torch
        import bitsandbytes as ref
        from oracle import oracle as O
        assert ref.__file__.endswith("__init__.pyc") and "bitsandbytes_amd" not in sys.modules
        W = torch.randn(64, 256)
        q, st = ref.functional.quantize_4bit(W, quant_type="nf4")
        q_o, am_o = O.quantize_4bit(W, 64, "nf4")
        assert torch.equal(q, q_o) and torch.equal(st.absmax, am_o)
        print("SOURCELESS_OK")
    """)
    assert "SOURCELESS_OK" in out


@pytest.mark.gpu
@pytest.mark.skipif(not compiled_reference_available(),
                    reason="oracle/_ref/ref_py (the byte-compiled reference package, built by oracle/build_ref.sh where /root/reference "
                           "exists) is not in this tree")
def test_mode_a_reference_host_code_over_this_library_on_gpu():
    out = _run("""
        os.environ["BNB_ROCM_VERSION"] = "70"      # cextension.py:36-47 -> libbitsandbytes_rocm70.so in the package directory
        farm = make_package(os.path.join(ROOT, "bitsandbytes_amd", "libbitsandbytes_mi355x.so"), "libbitsandbytes_rocm70.so")
        sys.path.insert(0, farm)
        import warnings
        import torch
        import bitsandbytes as ref
        from conftest import rel_err, same_values_ftz
        from oracle import oracle as O

        assert "bitsandbytes_amd" not in sys.modules
        lib = ref.cextension.lib
        assert type(lib).__name__ == "CudaBNBNativeLibrary" and lib.compiled_with_cuda, type(lib)
        assert "libbitsandbytes_mi355x.so" in open("/proc/self/maps").read()
        ops = sys.modules["bitsandbytes.backends.cuda.ops"]
        assert ops.lib is lib
        F = ref.functional
        dev = "cuda"
        torch.manual_seed(0)

        def oracle_weight(q_o, am, bs, qt, shape):
            return O.dequantize_4bit(q_o, am, bs, qt, shape, torch.float32).double()

        n_launch_checks = 0
        for qt, bs, dq, dtype in (("nf4", 64, False, torch.bfloat16), ("fp4", 128, True, torch.bfloat16),
                                  ("nf4", 64, False, torch.float16), ("nf4", 128, True, torch.float32),
                                  ("fp4", 64, False, torch.float32)):
            N, K = 768, 2048
            W = (torch.randn(N, K) / K**0.5).to(dtype)
            # ---- reference functional.quantize_4bit -> cquantize_blockwise_<T>_<qt> (+ the 8-bit pair for nested absmax)
            q, st = F.quantize_4bit(W.to(dev), blocksize=bs, quant_type=qt, compress_statistics=dq)
            q_o, am_o = O.quantize_4bit(W, bs, qt)
            assert torch.equal(q.cpu(), q_o), (qt, bs, dq, dtype)
            if dq:
                offset = st.offset.cpu()
                q8_o, am2_o = O.quantize_blockwise(am_o - offset, F.create_dynamic_map(), 256)
                assert torch.equal(st.absmax.cpu(), q8_o) and torch.equal(st.state2.absmax.cpu(), am2_o)
                am = O.dequantize_blockwise(q8_o, am2_o, F.create_dynamic_map(), 256, torch.float32) + offset
            else:
                assert torch.equal(st.absmax.cpu(), am_o)
                am = am_o
            # ---- reference functional.dequantize_4bit -> cdequantize_blockwise_<T>_<qt>
            d = F.dequantize_4bit(q, st)
            assert same_values_ftz(d.cpu(), O.dequantize_4bit(q_o, am, bs, qt, W.shape, dtype)), (qt, bs, dq, dtype)
            Wd = oracle_weight(q_o, am, bs, qt, W.shape)
            bias = torch.randn(N).to(dtype)
            tol = 1e-2 if dtype != torch.float32 else 2e-3
            for M in (1, 3, 4, 24, 64):
                x = torch.randn(M, K).to(dtype)
                y_ref = x.double() @ Wd.t()
                # reference autograd matmul_4bit -> bitsandbytes::gemm_4bit "cuda" kernel (backends/cuda/ops.py:921-982):
                # cgemm_4bit_<T> for M <= 4 (2 for fp32), the reference's dequantize + F.linear fallback above
                y = ref.matmul_4bit(x.to(dev), q.t(), st)
                assert y.shape == (M, N) and y.dtype == dtype
                assert rel_err(y.cpu(), y_ref) < tol, (qt, bs, dq, dtype, M, rel_err(y.cpu(), y_ref))
                yb = ref.matmul_4bit(x.to(dev), q.t(), st, bias=bias.to(dev))
                assert rel_err(yb.cpu(), y_ref + bias.double()) < tol
                # the reference's fused-kernel glue called directly at every M: this is what its dispatcher runs once a
                # maintainer raises the ROCm threshold for this library (INTEGRATION.md) - MFMA kernels for M > 4
                if st.nested:
                    y2 = ops._gemm_4bit_kernel_impl(x.to(dev), q, st.shape, st.state2.absmax, bs, qt, bias.to(dev), st.absmax,
                                                    st.state2.code, st.offset)
                else:
                    y2 = ops._gemm_4bit_kernel_impl(x.to(dev), q, st.shape, st.absmax, bs, qt, bias.to(dev), None, None, None)
                assert rel_err(y2.cpu(), y_ref + bias.double()) < tol, (qt, bs, dq, dtype, M)
                n_launch_checks += 3
            # ---- legacy functional.gemv_4bit -> cgemm_4bit_inference_naive_<T> (backends/cuda/ops.py:494-580)
            x = torch.randn(1, K).to(dtype)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                yv = F.gemv_4bit(x.to(dev), q.t(), state=st)
            assert rel_err(yv.cpu(), x.double() @ Wd.t()) < tol
            n_launch_checks += 1

        # ---- reference nn.Linear4bit: lazy quantization on .to("cuda"), forward, state-dict round trip
        lin = ref.nn.Linear4bit(1024, 512, bias=True, compute_dtype=torch.bfloat16, quant_type="nf4")
        W = lin.weight.data.clone()
        lin = lin.to(dev)
        assert lin.weight.dtype == torch.uint8 and lin.weight.quant_state is not None
        q_o, am_o = O.quantize_4bit(W, 64, "nf4")
        assert torch.equal(lin.weight.data.cpu().view(-1, 1), q_o)
        x = torch.randn(2, 5, 1024, dtype=torch.bfloat16)
        y = lin(x.to(dev))
        Wd = oracle_weight(q_o, am_o, 64, "nf4", (512, 1024))
        y_ref = x.double() @ Wd.t() + lin.bias.detach().cpu().double()
        assert y.shape == (2, 5, 512) and rel_err(y.detach().cpu(), y_ref) < 1e-2
        # the reference's own serialization recipe (tests/test_linear4bit.py:66-88)
        sd = lin.state_dict()
        bias2, weight2 = sd.pop("bias"), sd.pop("weight")
        lin2 = ref.nn.Linear4bit(1024, 512, bias=True, compute_dtype=torch.bfloat16, quant_type="nf4", device="meta")
        lin2.weight = ref.nn.Params4bit.from_prequantized(quantized_stats=sd, data=weight2, device=dev)
        lin2.bias = torch.nn.Parameter(bias2)
        lin2 = lin2.to(dev)
        assert torch.equal(lin2(x.to(dev)), y)
        # backward through the reference's MatMul4Bit (autograd/_functions.py:365-386): dequantize_4bit of this library + matmul
        xg = x.to(dev).requires_grad_(True)
        lin(xg).float().square().sum().backward()
        g_ref = (2 * y_ref) @ Wd
        assert rel_err(xg.grad.detach().cpu(), g_ref) < 2e-2
        torch.cuda.synchronize()
        print("MODE_A_GPU_OK", n_launch_checks)
    """)
    assert "MODE_A_GPU_OK" in out
