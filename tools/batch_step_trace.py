#!/usr/bin/env python3
"""Per STEP of a tools/batch_stall_probe.py run under rocprofv3 --kernel-trace --memory-copy-trace (IDLE_MS >= 5 puts a gap between the
steps): span, busy time of the uploads, of huffman_scan_kernel and of the reconstruction kernel, and the longest single upload --
what a slow step is slow IN.      python tools/batch_step_trace.py <trace dir> [gap ms = 3]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
gap = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 3e6
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mij::", "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "huff" if "huffman_scan" in name else "recon" if "fused" in name else "other"))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "HOST_TO_DEVICE" in r["Direction"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 50_000:
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "h2d"))
ev.sort()
steps, cur, last_end = [], [], None
for e in ev:
    if last_end is not None and e[0] - last_end > gap and cur:
        steps.append(cur)
        cur = []
    cur.append(e)
    last_end = max(last_end or 0, e[1])
if cur:
    steps.append(cur)
print(f"{len(steps)} steps (activity separated by > {gap / 1e6:.1f} ms)")
print("  step   span ms   h2d busy   h2d n   longest h2d   huff busy  huff n  mean huff   recon busy  mean recon")
for i, s in enumerate(steps):
    if len(s) < 4:
        continue
    span = (max(e[1] for e in s) - s[0][0]) / 1e6
    def agg(k):
        xs = [(e[1] - e[0]) / 1e6 for e in s if e[2] == k]
        return sum(xs), len(xs), (max(xs) if xs else 0.0)
    hb, hn, hl = agg("h2d")
    fb, fn, _ = agg("huff")
    rb, rn, _ = agg("recon")
    print(f"  {i:4d}  {span:8.2f}  {hb:9.2f}  {hn:5d}  {hl:11.3f}  {fb:10.2f}  {fn:6d}  {fb / max(1, fn):9.3f}  {rb:10.2f}  {rb / max(1, rn):10.3f}")
