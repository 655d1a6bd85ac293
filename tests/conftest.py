import os
import random
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a host without a GPU skips the GPU tests instead of erroring; on a GPU box they
    run (and `BNB_REQUIRE_GPU=1` turns a missing GPU into a hard failure there)."""
    if torch.cuda.is_available() or os.environ.get("BNB_REQUIRE_GPU") == "1":
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _seed_everything():
    # same policy as the reference's tests/conftest.py:9-29
    torch.manual_seed(0)
    np.random.seed(0)
    random.seed(0)
    yield


_GOLDEN = None
_NP2T = {0: (np.int32, torch.float32), 1: (np.int16, torch.float16), 2: (np.int16, torch.bfloat16)}
DT = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}
QT = {1: "fp4", 2: "nf4"}


def golden():
    global _GOLDEN
    if _GOLDEN is None:
        _GOLDEN = np.load(os.path.join(ROOT, "tests", "golden", "golden_4bit.npz"))
    return _GOLDEN


_GOLDEN_M8 = None


def golden_8bit_maps():
    """tests/golden/golden_8bit_maps.npz (make_golden_8bit_maps.py): the 8-bit op on the other code maps."""
    global _GOLDEN_M8
    if _GOLDEN_M8 is None:
        _GOLDEN_M8 = np.load(os.path.join(ROOT, "tests", "golden", "golden_8bit_maps.npz"))
    return _GOLDEN_M8


def from_bits(arr: np.ndarray, dtype_code: int) -> torch.Tensor:
    """Inverse of make_golden.bits()."""
    t = torch.from_numpy(np.ascontiguousarray(arr))
    return t.view(DT[dtype_code])


def same_values(a: torch.Tensor, b: torch.Tensor) -> bool:
    """Bit-for-bit up to the sign of zero (the reference's own CPU code paths disagree on -0 vs +0
    for FP4 code 8: csrc/cpu_ops.cpp:271-276 vs :279-282)."""
    return a.shape == b.shape and a.dtype == b.dtype and bool(torch.equal(a.float(), b.float()))


def same_values_ftz(a: torch.Tensor, b: torch.Tensor) -> bool:
    """same_values, with results below the smallest normal fp32 treated as zero. Needed only for bf16
    outputs: the reference's AVX512-BF16 convert (vcvtneps2bf16, csrc/cpu_ops.cpp:386) flushes
    denormals to zero while its scalar path (and a GPU) keeps them."""
    fa, fb = a.float(), b.float()
    tiny = 2.0**-126
    fa = torch.where(fa.abs() < tiny, torch.zeros_like(fa), fa)
    fb = torch.where(fb.abs() < tiny, torch.zeros_like(fb), fb)
    return a.shape == b.shape and a.dtype == b.dtype and bool(torch.equal(fa, fb))


def rel_err(y: torch.Tensor, ref: torch.Tensor) -> float:
    y, ref = y.double().flatten(), ref.double().flatten()
    return float((y - ref).norm() / ref.norm().clamp_min(1e-30))


def gpu_ready() -> bool:
    return torch.cuda.is_available()
