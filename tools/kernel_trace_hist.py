#!/usr/bin/env python3
"""Histogram of per-dispatch kernel durations (and start-to-start gaps) from a rocprofv3 --kernel-trace CSV.
    python tools/kernel_trace_hist.py <kernel_trace.csv> [name-substring]"""
import csv
import sys

import numpy as np

path = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "gemv4_stream"
rows = []
with open(path, newline="") as fh:
    for r in csv.DictReader(fh):
        if sub in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort()
st = np.array([r[0] for r in rows], dtype=np.int64)
en = np.array([r[1] for r in rows], dtype=np.int64)
dur = (en - st) / 1e3
gap = (st[1:] - st[:-1]) / 1e3
idle = (st[1:] - en[:-1]) / 1e3
print(f"{len(rows)} dispatches of *{sub}*")
for name, a in (("duration us", dur), ("start-to-start us", gap), ("end-to-next-start us", idle)):
    q = np.percentile(a, [0, 10, 50, 90, 99, 100])
    print(f"{name:22s} mean {a.mean():7.2f}  min {q[0]:6.2f}  p10 {q[1]:6.2f}  p50 {q[2]:6.2f}  p90 {q[3]:6.2f}  p99 {q[4]:6.2f}  max {q[5]:7.2f}")
h, edges = np.histogram(dur, bins=np.arange(3.0, 10.5, 0.5))
print("duration histogram (us): " + "  ".join(f"{edges[i]:.1f}:{h[i]}" for i in range(len(h)) if h[i]))
print("first 40 durations:", " ".join(f"{d:.1f}" for d in dur[:40]))
k = 128
if len(dur) >= 2 * k:
    m = dur[: len(dur) // k * k].reshape(-1, k)
    print("mean duration by position in the 128-layer step (first 16):", " ".join(f"{v:.1f}" for v in m.mean(axis=0)[:16]))
    print("mean duration by replay:", " ".join(f"{v:.2f}" for v in m.mean(axis=1)))
