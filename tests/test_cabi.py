"""CPU: the C-ABI library builds, loads and exports every symbol include/bnb_mi355x.h declares.
No compute calls (there is no GPU here)."""
import ctypes as ct
import os
import re

from conftest import ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "bnb_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:void\*?|void|int|size_t|const char\*)\s+\*?([A-Za-z_][A-Za-z0-9_]*)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_header_declares_reference_abi():
    names = _header_symbols()
    for d in ("fp32", "fp16", "bf16"):
        assert f"cgemm_4bit_{d}" in names
        assert f"cgemm_4bit_inference_naive_{d}" in names
        assert f"cquantize_blockwise_{d}" in names and f"cdequantize_blockwise_{d}" in names
        for q in ("nf4", "fp4"):
            assert f"cquantize_blockwise_{d}_{q}" in names and f"cdequantize_blockwise_{d}_{q}" in names
    assert "get_context" in names and "cget_managed_ptr" in names


def test_library_exports_every_declared_symbol():
    from bitsandbytes_amd import cextension as ce

    assert ce.lib, f"{ce.LIB_PATH} not built: run __graft_entry__.build() / make -C bitsandbytes_amd/csrc"
    dll = ct.CDLL(str(ce.LIB_PATH))
    missing = [n for n in _header_symbols() if not hasattr(dll, n)]
    assert not missing, f"declared in include/bnb_mi355x.h but not exported: {missing}"
    assert set(ce.EXPORTED_SYMBOLS) <= set(_header_symbols())
    assert ce.lib.bnb_mi355x_version().decode().endswith("gfx950")


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    """The product path must raise, never fall back, when the HIP library is absent."""
    import importlib

    import pytest

    from bitsandbytes_amd import cextension as ce

    stub = ce.MissingNativeLibrary("simulated")
    assert not stub
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        stub.cgemm_4bit_bf16
    monkeypatch.setenv("BNB_MI355X_LIBRARY", str(tmp_path / "nope.so"))
    ce2 = importlib.reload(ce)
    try:
        assert isinstance(ce2.lib, ce2.MissingNativeLibrary)
    finally:
        monkeypatch.delenv("BNB_MI355X_LIBRARY")
        importlib.reload(ce)


def test_no_cpu_kernels_registered_by_product():
    """Calling an op on CPU tensors must fail (no CPU kernel) unless a test registered the oracle."""
    import subprocess
    import sys

    code = (
        "import torch, bitsandbytes_amd as b\n"
        "try:\n"
        "    torch.ops.bitsandbytes.quantize_4bit.default(torch.randn(64), 64, 'nf4', torch.uint8)\n"
        "    print('COMPUTED')\n"
        "except NotImplementedError:\n"
        "    print('RAISED')\n"
    )
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT).stdout
    assert "RAISED" in out and "COMPUTED" not in out
