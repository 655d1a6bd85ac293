"""CPU: the REFERENCE's own test files, executed unmodified and in place (from /root/reference/tests) against
this package's host layer: ``import bitsandbytes`` resolves to ``bitsandbytes_amd`` and the CPU arithmetic is
the oracle's (tests/_reference_suite_shim.py). This is the broadest drop-in check available without a GPU:
module construction, lazy quantization, dtype policy, (de)serialisation, pickling/copying, FSDP quant-state
recovery, parametrization hooks and state-dict layout, op schemas / opcheck and torch.compile tracing, all as
the reference's maintainers wrote the expectations. Skipped where the reference checkout is absent (GPU box).
Set BNB_COMPAT_FULL=1 for the long selections (the 4-bit functional class and the torch.compile matrix)."""
import os
import subprocess
import sys

import pytest

REF = os.environ.get("BNB_REFERENCE_DIR", "/root/reference")
FULL = os.environ.get("BNB_COMPAT_FULL", "0") == "1"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "tests")), reason="reference checkout not present")

# (file, -k expression, minimum number of tests that must pass)
SELECTIONS = [
    ("tests/test_linear4bit.py", "not compile and not fsdp", 170),
    ("tests/test_parametrize.py", "", 80),
    ("tests/test_autograd.py", "matmul_4bit", 40),
    ("tests/test_ops.py", "4bit", 250),
    ("tests/test_modules.py", "(embedding or 4bit or NF4 or FP4) and not 8bit and not Int8 and not int8", 20),
    # default: everything 4-bit except the long gemv / large-tensor sweeps (they dominate the runtime on CPU)
    ("tests/test_functional.py", "4bit" if FULL else "4bit and not benchmark and not test_gemv_4bit and not quant_large", 60),
    # the general 8-bit blockwise op: dynamic / linear / fp8 code maps, QuantState round trip
    ("tests/test_functional.py", "Test8BitBlockwiseQuantizeFunctional and not bench", 8),
]
if FULL:
    SELECTIONS.append(("tests/test_linear4bit.py", "compile", 100))


@pytest.fixture(scope="module")
def shim_root():
    import _reference_suite_shim

    return _reference_suite_shim.build(REF)


@pytest.mark.parametrize("path,expr,min_passed", SELECTIONS, ids=[f"{s[0].split('/')[-1]}[{s[1] or 'all'}]" for s in SELECTIONS])
def test_reference_test_file_passes_against_this_package(shim_root, path, expr, min_passed):
    env = dict(os.environ, BNB_TEST_DEVICE="cpu", PYTHONPATH=shim_root, PYTHONDONTWRITEBYTECODE="1")
    cmd = [sys.executable, "-m", "pytest", path, "-q", "-p", "no:cacheprovider", "-x"]
    if expr:
        cmd += ["-k", expr]
    proc = subprocess.run(cmd, cwd=shim_root, env=env, capture_output=True, text=True, timeout=3000)
    tail = "\n".join(proc.stdout.splitlines()[-15:])
    assert proc.returncode == 0, f"{path} -k '{expr}' failed:\n{tail}\n{proc.stderr[-1500:]}"
    summary = proc.stdout.strip().splitlines()[-1]
    passed = int(summary.split(" passed")[0].split()[-1])
    assert passed >= min_passed, summary
