#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs written by tools/gpu_profile.sh into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(out, sub, "**", pat), recursive=True))


for f in find("trace", "*kernel_stats.csv"):
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, out))
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i < 8:
                print("  " + ",".join(row))
for sub in ("pmc_sq", "pmc_lds", "pmc_fetch", "pmc_write"):
    for f in find(sub, "*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        with open(f) as fh:
            for row in csv.DictReader(fh):
                acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print(f"== counters ({sub}): per-dispatch mean")
        for k, ctrs in acc.items():
            if not any(t in k for t in ("fused420", "fused444", "idct_planes", "upsample_color", "huffman", "xt_merge")):
                continue
            print("  kernel", k)
            for c, v in sorted(ctrs.items()):
                print(f"    {c:28s} n={len(v):3d} mean={sum(v) / len(v):.6g}")
