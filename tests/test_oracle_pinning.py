"""CPU: pins the oracle against the LIVE reference (authoring container only; skipped where
/root/reference or oracle/_ref is absent, e.g. on the GPU box). The actual checks are in
tests/pinning_impl.py and run in a separate interpreter, because the reference package must define
the ``bitsandbytes::`` operator schemas before bitsandbytes_amd is imported."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from oracle.ref_import import reference_available


@pytest.mark.skipif(not reference_available(), reason="needs /root/reference and oracle/_ref (oracle/build_ref.sh)")
def test_oracle_pinned_against_live_reference():
    proc = subprocess.run(
        [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "pinning_impl.py"), "-x", "-q", "-p", "no:cacheprovider"],
        capture_output=True, text=True, cwd=ROOT, timeout=900,
    )
    tail = "\n".join(proc.stdout.splitlines()[-25:])
    assert proc.returncode == 0, f"pinning suite failed:\n{tail}\n{proc.stderr[-2000:]}"
    assert " passed" in tail and "skipped" not in tail.split("passed")[0][-40:], tail


@pytest.mark.skipif(not reference_available(), reason="needs /root/reference and oracle/_ref (oracle/build_ref.sh)")
def test_expected_failure_list_of_the_8bit_envelopes_is_pinned_by_the_reference_cpu_backend():
    """The ids of tests/golden/reference_suite_expected_failures.txt that belong to the reference's 8-bit envelope test: the reference's
    OWN CPU backend misses the same envelopes (tests/pinning_8bit_envelope_impl.py, its own interpreter)."""
    proc = subprocess.run(
        [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "pinning_8bit_envelope_impl.py"), "-x", "-q", "-p", "no:cacheprovider"],
        capture_output=True, text=True, cwd=ROOT, timeout=900,
    )
    tail = "\n".join(proc.stdout.splitlines()[-25:])
    assert proc.returncode == 0, f"8-bit envelope pinning failed:\n{tail}\n{proc.stderr[-2000:]}"
    assert "3 passed" in tail, tail
