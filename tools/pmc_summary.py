#!/usr/bin/env python3
"""Average the counters of one kernel from a rocprofv3 counter_collection CSV.
    python tools/pmc_summary.py <dir> <kernel substring>"""
import csv
import glob
import os
import sys
from collections import defaultdict

d, sub = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            if sub in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    v = v[len(v) // 3:]  # drop the first (cold) round
    print(f"{k:32s} mean {sum(v) / len(v):16.1f}  over {len(v)} dispatches")
