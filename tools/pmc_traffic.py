#!/usr/bin/env python3
"""HBM-side traffic per launch of the fused-op kernels at one shape, from rocprofv3 PMC counters - the recipe of
/opt/skills/guides/MI355X_MICROARCH.md (HBM section) that bench.py uses for the gemv: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
(they do not fit one TCC pass), nothing beside them but the kernel trace; both report KiB; on gfx950 FETCH_SIZE tallies the 128-byte
requests of wide coalesced reads at 64 bytes, so it is doubled. WRITE_SIZE is uncalibrated on gfx950 (guide): this tool prints it
next to byte counts known BY CONSTRUCTION (the split-K slabs a kernel writes, the output the finalize launch writes), which is the
calibration the guide asks for.
    python tools/pmc_traffic.py --n 8192 --k 8192 --m 64 [--knob 4000] [--layers 8]
Prints one line per kernel family seen (gemm4_* / gemv4_*): launches, 2 x FETCH_SIZE and WRITE_SIZE in MB per launch, and the
algorithmic bytes of the op with the slab bytes the plan implies."""
import argparse
import csv
import glob
import os
import shutil
import subprocess
import sys
import tempfile
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    for fam in ("gemm4_finalize_kq_kernel", "gemm4_finalize_kernel", "gemm4_mfma_kq_kernel", "gemm4_mfma_pc_kernel", "gemm4_mfma_rt_kernel",
                "gemm4_mfma_sm_kernel", "gemm4_mfma_tall_kernel",
                "gemv4_stream_kernel", "dequantize4_kernel", "quantize4_kernel"):
        if fam in name:
            return fam
    return None


def one_pass(counter, child_args, timeout_s):
    exe = shutil.which("rocprofv3")
    if exe is None:
        raise SystemExit("rocprofv3 not on PATH")
    out_dir = tempfile.mkdtemp(prefix="bnb_pmc_", dir="/tmp")
    cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out_dir, "-o", "pmc", "--",
           sys.executable, os.path.join(ROOT, "tools", "pmc_mfma.py")] + child_args
    env = dict(os.environ, TMPDIR="/tmp")
    acc = defaultdict(list)
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
        for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for r in csv.DictReader(fh):
                    fam = short(r.get("Kernel_Name", ""))
                    if fam and r.get("Counter_Name") == counter:
                        acc[fam].append(float(r["Counter_Value"]))
    finally:
        shutil.rmtree(out_dir, ignore_errors=True)
    # (pmc_mfma.py makes three sweeps over its layers: the first one is cold - allocations, first touch)
    return {k: (sum(v[len(v) // 3:]) / len(v[len(v) // 3:]), len(v)) for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--k", type=int, default=8192)
    ap.add_argument("--m", type=int, default=64)
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--knob", type=int, default=0)
    ap.add_argument("--timeout", type=int, default=240)
    a = ap.parse_args()
    child = ["--n", str(a.n), "--k", str(a.k), "--m", str(a.m), "--layers", str(a.layers), "--knob", str(a.knob)]
    fetch = one_pass("FETCH_SIZE", child, a.timeout)
    write = one_pass("WRITE_SIZE", child, a.timeout)
    alg = a.n * a.k // 2 + 4 * a.n * a.k // 64 + 2 * a.m * a.k + 2 * a.m * a.n
    print(f"# {a.n} x {a.k}, M = {a.m}, knob {a.knob}: algorithmic bytes of the op {alg / 1e6:.2f} MB (weights {a.n * a.k / 2e6:.2f} + absmax "
          f"{4 * a.n * a.k / 64 / 1e6:.2f} + A {2 * a.m * a.k / 1e6:.2f} + out {2 * a.m * a.n / 1e6:.2f}); one fp32 slab [M, N] = {4 * a.m * a.n / 1e6:.2f} MB")
    print(f"{'kernel':28s} {'launches':>8s} {'2 x FETCH_SIZE MB':>18s} {'WRITE_SIZE MB':>14s}")
    tot_f = tot_w = 0.0
    for fam in sorted(set(fetch) | set(write)):
        f = 2.0 * fetch.get(fam, (0.0, 0))[0] * 1024 / 1e6
        w = write.get(fam, (0.0, 0))[0] * 1024 / 1e6
        setup = fam in ("quantize4_kernel", "dequantize4_kernel")  # (the tool quantizes its layers: not part of the op)
        if not setup:
            tot_f, tot_w = tot_f + f, tot_w + w
        print(f"{fam:28s} {fetch.get(fam, (0, 0))[1]:8d} {f:18.2f} {w:14.2f}" + ("   (set-up of the tool, not in the sum)" if setup else ""))
    print(f"{'sum over the op':28s} {'':8s} {tot_f:18.2f} {tot_w:14.2f}   = {(tot_f + tot_w) / (alg / 1e6):.2f} x the algorithmic bytes")


if __name__ == "__main__":
    main()
