"""GPU (-m gpu): the path's bit-for-bit identities on MANY random inputs (tests/checks/identity_stress.py, its own process: it opens
a one-rank process group for the peer chain). The other tests check each identity on a handful of inputs; a one-fma / two-roundings
difference in the double-quantised scale survived two rounds of those by luck."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import gpu_ready

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_identities_hold_on_many_random_inputs():
    if not gpu_ready() and not os.path.exists("/dev/kfd") and os.environ.get("BNB_REQUIRE_GPU") != "1":
        pytest.skip("no GPU device on this host")
    assert gpu_ready(), "GPU tests selected but torch.cuda.is_available() is False"
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "checks", "identity_stress.py"), "--iters", "36"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "IDENTITY_STRESS OK" in p.stdout, p.stdout[-3000:]


def test_every_kernel_is_deterministic_run_to_run_under_a_busy_memory_system():
    """tests/checks/determinism_stress.py: ~200 cases (routed and forced geometries of every kernel family, forward and backward, the
    standalone kernels), each repeated while a second stream keeps copying: every result equals the first, bit for bit."""
    if not gpu_ready() and not os.path.exists("/dev/kfd") and os.environ.get("BNB_REQUIRE_GPU") != "1":
        pytest.skip("no GPU device on this host")
    assert gpu_ready(), "GPU tests selected but torch.cuda.is_available() is False"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "checks", "determinism_stress.py"), "--runs", "60"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0 and "DETERMINISM_STRESS OK" in p.stdout, p.stdout[-3000:]


def test_random_shapes_through_the_public_operators_against_the_oracle():
    """tests/checks/fuzz_vs_oracle.py (imports the oracle: test infrastructure): 120 random draws of shape / blocksize / type / dtype /
    statistics kind / bias - quantize bit-exact, dequantize equal, gemm_4bit within tolerance of fp32 dequantize + fp32 linear."""
    if not gpu_ready() and not os.path.exists("/dev/kfd") and os.environ.get("BNB_REQUIRE_GPU") != "1":
        pytest.skip("no GPU device on this host")
    assert gpu_ready(), "GPU tests selected but torch.cuda.is_available() is False"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "checks", "fuzz_vs_oracle.py"), "--draws", "120", "--seed", "7"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0 and "FUZZ OK" in p.stdout, p.stdout[-3000:]


def test_no_entry_point_writes_outside_its_outputs_or_needs_more_than_element_alignment():
    """tests/checks/guard_stress.py: every operand of every C-ABI entry point carved out of a larger buffer at a random element
    offset, outputs and scratch between guard bands: bands untouched, results equal to the public operator on aligned tensors."""
    if not gpu_ready() and not os.path.exists("/dev/kfd") and os.environ.get("BNB_REQUIRE_GPU") != "1":
        pytest.skip("no GPU device on this host")
    assert gpu_ready(), "GPU tests selected but torch.cuda.is_available() is False"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "checks", "guard_stress.py"), "--draws", "60", "--seed", "5"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert p.returncode == 0 and "GUARD_STRESS OK" in p.stdout, p.stdout[-3000:]
