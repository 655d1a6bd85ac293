"""CPU restatement (numpy float32) of the two threshold finders of csrc/blockwise8.hip and of the byte-table construction.

T(m) = min{u in [0, 65536] : bin_value(u) > m},  bin_value(u) = -1 + (2 u) / 65535 in fp32 with IEEE division. The kernels used
to find it by 17 bisection steps; they now start from the algebraic inverse and walk (first_bin_above). The byte table of the
large-input encoder is lut[u] = #{i : T_i <= u}, built on the device as scatter + prefix sum. This script checks, for every
code map the reference can construct, that both finders give the same thresholds and that the prefix-sum construction equals
the direct count (and the reference's own table rule, csrc/cpu_ops.cpp:501-520).

    python tests/checks/emulate_q8_thresholds.py        (also imported by tests/test_host_logic.py)
"""
import numpy as np

F32 = np.float32


def bin_value(u):
    return F32(-1.0) + (F32(2.0) * F32(u)) / F32(65535.0)


def bisect_threshold(m):
    lo, hi = 0, 65536
    for _ in range(17):
        if lo >= hi:
            break
        c = (lo + hi) >> 1
        if bin_value(c) > m:
            hi = c
        else:
            lo = c + 1
    return hi


def first_bin_above(m):
    m = F32(m)
    if m < F32(-1.0):
        return 0
    if not (m < F32(1.0)):
        return 65536
    c = int((m + F32(1.0)) * F32(32767.5))
    c = min(max(c, 0), 65535)
    while c > 0 and bin_value(c - 1) > m:
        c -= 1
    while c < 65536 and not (bin_value(c) > m):
        c += 1
    return c


def check_code(code):
    code = np.asarray(code, dtype=np.float32)
    assert code.shape == (256,)
    mid = [F32(0.5) * (code[i] + code[i + 1]) for i in range(255)] + [F32(np.inf)]
    thr = np.array([first_bin_above(m) for m in mid], dtype=np.int64)
    assert np.array_equal(thr, np.array([bisect_threshold(m) for m in mid], dtype=np.int64))
    # byte table as the device builds it: marks + inclusive prefix sum
    marks = np.zeros(65536, dtype=np.int64)
    for t in thr[:255]:
        if t < 65536:
            marks[t] += 1
    lut = np.cumsum(marks)
    assert lut.max() <= 255
    # ... equals the direct count, and the reference's rule lut[u] = #{i : mid_i < bin_value(u)}
    u = np.arange(65536)
    direct = (thr[None, :255] <= u[:, None]).sum(axis=1)
    assert np.array_equal(lut, direct)
    vals = (F32(-1.0) + (F32(2.0) * u.astype(np.float32)) / F32(65535.0)).astype(np.float32)
    ref = (np.array(mid[:255], dtype=np.float32)[None, :] < vals[:, None]).sum(axis=1)
    assert np.array_equal(lut, ref)
    return True


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    assert check_code(np.sort(rng.uniform(-1, 1, 256)).astype(np.float32))
    assert check_code(np.linspace(-1, 1, 256, dtype=np.float32))
    for m in (-2.0, -1.0, np.nextafter(F32(-1.0), F32(0)), 0.0, np.nextafter(F32(1.0), F32(0)), 1.0, 2.0, np.inf, np.nan):
        assert first_bin_above(m) == bisect_threshold(F32(m)), m
    print("ok")
