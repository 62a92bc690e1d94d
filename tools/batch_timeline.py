#!/usr/bin/env python3
"""Per-engine timeline of the LAST batch in a rocprofv3 --kernel-trace --memory-copy-trace run of tools/batch4k_bench.py:
   python tools/batch_timeline.py <dir with t_kernel_trace.csv and t_memory_copy_trace.csv> [chunks per batch]
Prints start / end / duration (ms from the batch's first upload), what ran (copy direction or kernel), the queue / stream id, and the
busy time of each engine.  The batch = the last `chunks per batch` large uploads and everything from the first of them on."""
import csv
import glob
import os
import sys

d = sys.argv[1]
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("mij::", "")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name[:28], r.get("Queue_Id", r.get("Stream_Id", "?"))))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        kind = r["Direction"].replace("MEMORY_COPY_", "").replace("HOST_TO_DEVICE", "H2D").replace("DEVICE_TO_HOST", "D2H").replace("DEVICE_TO_DEVICE", "D2D")
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, r.get("Stream_Id", "?")))
ev.sort()
big = [e for e in ev if e[2] == "H2D" and e[1] - e[0] > 100_000]  # the chunks' uploads (not the table blobs and status words)
nchunks = int(sys.argv[2]) if len(sys.argv) > 2 else 11           # uploads per batch: ceil(frames / chunk)
t0 = big[-nchunks][0]
last = [e for e in ev if e[0] >= t0 - 1000]
t_end = max(e[1] for e in last)
print(f"last batch: {len(last)} events, {(t_end - t0) / 1e6:.3f} ms from its first upload to the end of its last event")
print("  start      end      dur   what                          queue")
busy = {}
for s, e, what, q in last:
    if e - s < 3000 and what in ("H2D", "D2D"):
        continue  # status words and table blobs
    print(f"{(s - t0) / 1e6:8.3f} {(e - t0) / 1e6:8.3f} {(e - s) / 1e6:7.3f}   {what:<28s}  {q}")
    busy[what] = busy.get(what, 0) + (e - s)
print("busy ms:", {k: round(v / 1e6, 3) for k, v in sorted(busy.items())})
