#!/bin/bash
# kernel + copy timeline of the last no-DRI decode of tools/nodri_timeline.py
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
python $R/tools/nodri_timeline.py
rm -rf /tmp/nd; timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/nd -o t -- python $R/tools/nodri_timeline.py > /tmp/nd.log 2>&1
grep "ms" /tmp/nd.log | tail -2
python - <<PY
import csv, glob
ev = []
for f in glob.glob("/tmp/nd/**/t_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-44:]))
for f in glob.glob("/tmp/nd/**/t_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
# the last decode: from the last upload of a big buffer on
big = [i for i, e in enumerate(ev) if e[2].startswith("copy") and e[2].split()[-1].isdigit() and int(e[2].split()[-1]) > 2000000]
i0 = big[-1] if big else 0
t0 = ev[i0][0]
for s, e, n in ev[i0:]:
    print("%9.1f us  +%8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
