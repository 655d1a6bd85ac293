"""(Run through tests/test_oracle_pinning.py, in its own process - the reference package must define the ``bitsandbytes::``
operator schemas first.)

CPU, authoring container only. tests/golden/reference_suite_expected_failures.txt lists 68 ids of the reference's
``Test8BitBlockwiseQuantizeFunctional::test_dynamic_blockwise_quantization`` (/root/reference/tests/test_functional.py:105-169) that
fail when the reference's own suite runs against this package on the HIP device. The claim behind the list: those error envelopes are
calibrated on the CUDA kernels' quantization rule, this package reproduces the CPU backend's rule bit for bit, and under THAT rule
the reference itself misses them. Here the claim is put to the reference: the body of that test runs on device "cpu" with the
reference's own package and its own libbitsandbytes_cpu.so (the reference skips these parameters on the CPU only because they are
slow there), and the reference must miss the envelopes for the listed ids.

What is asserted, and why in this form:
 * fp16 / bf16 (56 ids, all of them listed): the reference CPU backend misses the relative-error threshold by more than 10x, every
   blocksize, nested or not, both code maps. (The statistic divides by |A1.float() + 1e-8|: a half-precision input that is exactly
   zero contributes ~1e8 x its reconstruction.)
 * fp32 (28 ids, the 12 with blocksize >= 1024 listed): the statistic is mean(|A1 - A2| / |A1 + 1e-8|) over a normal sample - a
   floor set by the code's resolution, which grows with the blocksize (a bigger block has a bigger absmax), plus a heavy tail from the
   samples next to zero (the CPU rule sends them to the smallest non-zero code). Single iterations of the SAME parameters range from
   0.0164 to 0.0297 around a threshold of 0.018, so the mean over a few CPU iterations is a coin flip and is NOT asserted - for
   these ids this test pins how close the reference's own backend sits to the threshold, not a guaranteed miss. The floor - the
   MINIMUM over the iterations, the draw with the fewest samples next to zero - is stable to 1e-4 across seeds: it is asserted to
   lie above 90 % of the threshold exactly for the blocksizes the list names (>= 1024: 0.0164, 0.0168, 0.0171), below it for the
   others (<= 512: 0.0143 ... 0.0160), and to grow with the blocksize; the tail then adds 0.001 - 0.002 on top. The absolute-error
   envelopes hold everywhere. (On the HIP device the 12 listed ids have failed in every run of the reference suite so far, the 16
   others passed: tests/test_gpu_reference_suite.py pins both.)
 * the list's 8-bit part is exactly {all half-precision ids} + {fp32 ids with blocksize >= 1024}."""
import itertools
import os
import re
import statistics

import pytest
import torch

from oracle.ref_import import reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason="needs /root/reference and oracle/_ref (build_ref.sh)")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOCKSIZES = (4096, 2048, 1024, 512, 256, 128, 64)
DTYPES = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}


def _expected_ids():
    ids = set()
    with open(os.path.join(ROOT, "tests", "golden", "reference_suite_expected_failures.txt")) as fh:
        for line in fh:
            m = re.search(r"Test8BitBlockwiseQuantizeFunctional::test_dynamic_blockwise_quantization\[(.*)-cuda\]", line)
            if m:
                ids.add(m.group(1))
    return ids


def _tuple_id(signed, bs, nested, dname):
    return f"signed={'T' if signed else 'F'}-{bs}-nested={'T' if nested else 'F'}-{dname}"


@pytest.fixture(scope="module")
def F():
    from oracle.ref_import import import_reference

    import_reference()
    import bitsandbytes.functional as F

    return F


def _envelope(F, dtype, nested, bs, signed, iters):
    """The statistics of the reference test's two parts (test_functional.py:128-166), per iteration."""
    d1, r1, d2, r2 = [], [], [], []
    for _ in range(iters):
        A1 = torch.randn(1024, 1024, dtype=dtype)
        C, S = F.quantize_blockwise(A1, blocksize=bs, nested=nested)
        A2 = F.dequantize_blockwise(C, S)
        diff = torch.abs(A1 - A2).float()
        d1.append(diff.mean().item())
        r1.append((diff / torch.abs(A1.float() + 1e-8)).mean().item())
    code = F.create_dynamic_map(signed=signed)
    for _ in range(iters):
        A1 = torch.rand(1024, 1024, dtype=dtype)
        C, S = F.quantize_blockwise(A1, blocksize=bs, nested=nested, code=code)
        A2 = F.dequantize_blockwise(C, S)
        diff = torch.abs(A1 - A2).float()
        d2.append(diff.mean().item())
        r2.append((diff / torch.abs(A1.float() + 1e-8)).mean().item())
    return d1, r1, d2, r2


def test_the_list_is_half_precision_plus_fp32_at_large_blocks():
    want = {_tuple_id(s, bs, n, d) for s, bs, n, d in itertools.product((False, True), BLOCKSIZES, (False, True), ("fp16", "bf16"))}
    want |= {_tuple_id(s, bs, n, "fp32") for s, bs, n in itertools.product((False, True), (4096, 2048, 1024), (False, True))}
    assert _expected_ids() == want and len(want) == 68


def test_reference_cpu_backend_misses_the_half_precision_envelopes_by_an_order_of_magnitude(F):
    torch.manual_seed(0)
    listed = _expected_ids()
    for signed, bs, nested, dname in itertools.product((False, True), BLOCKSIZES, (False, True), ("fp16", "bf16")):
        assert _tuple_id(signed, bs, nested, dname) in listed
        _, r1, _, r2 = _envelope(F, DTYPES[dname], nested, bs, signed, 2)
        # thresholds of the reference test: relerr < 0.018 (normal data), < 0.015 / 0.012 (signed / unsigned dynamic map)
        assert statistics.mean(r1) > 10 * 0.018, (signed, bs, nested, dname, r1)
        assert statistics.mean(r2) > 10 * 0.015, (signed, bs, nested, dname, r2)


def test_reference_cpu_backend_fp32_envelope_floor_crosses_the_threshold_where_the_list_says(F):
    torch.manual_seed(0)
    listed = _expected_ids()
    iters = 10  # the reference's own iteration count off the accelerators (test_functional.py:117-118)
    for nested in (False, True):
        floors = []
        for bs in sorted(BLOCKSIZES):
            d1, r1, d2, r2 = _envelope(F, torch.float32, nested, bs, True, iters)
            # absolute errors: inside the envelope everywhere (0.011 normal data; 0.0036 signed dynamic map), so is the uniform part's relerr
            assert statistics.mean(d1) < 0.011 and statistics.mean(d2) < 0.0036 and statistics.mean(r2) < 0.015, (bs, nested)
            floor = min(r1)
            floors.append(floor)
            in_list = all(_tuple_id(s, bs, nested, "fp32") in listed for s in (False, True))
            none_in_list = not any(_tuple_id(s, bs, nested, "fp32") in listed for s in (False, True))
            assert in_list or none_in_list  # (the first part of the test does not depend on `signed`)
            assert (floor > 0.9 * 0.018) == in_list, (bs, nested, floor, in_list)
            assert max(r1) < 0.06 and min(r1) > 0.012, (bs, nested, r1)  # (the tail is heavy, not broken)
        assert floors == sorted(floors), (nested, floors)  # grows with the blocksize
