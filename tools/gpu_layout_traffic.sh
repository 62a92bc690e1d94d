#!/bin/bash
# HBM traffic of the reconstruction kernel per sampling layout (tools/layout_bench.py): rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in
# separate passes, FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md, against the algorithmic bytes of a launch.
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export LAYOUTS=${LAYOUTS:-cmyk,3x1,1x4,lumasub,3x3,444_12,422_12}
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr_$c
  timeout 500 rocprofv3 --pmc $c --output-format csv -d /tmp/tr_$c -o t -- python $R/tools/layout_bench.py > /tmp/tr_$c.log 2>&1
done
grep "ms/launch" /tmp/tr_FETCH_SIZE.log | cut -c1-10,60-175
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/tr_%s/**/t_counter_collection.csv" % c, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c and "copyBuffer" not in r["Kernel_Name"] and "elementwise" not in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:58], r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))][c].append(float(r["Counter_Value"]))
print("kernel, grid, LDS: launches, FETCH_SIZE KB (as counted), WRITE_SIZE KB, traffic MB = (2 FETCH + WRITE) / 1024")
for k, v in acc.items():
    if len(v["FETCH_SIZE"]) < 10: continue
    fe, wr = sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"]), sum(v["WRITE_SIZE"]) / max(len(v["WRITE_SIZE"]), 1)
    print(k, len(v["FETCH_SIZE"]), "%.0f %.0f -> %.1f MB per launch" % (fe, wr, (2 * fe + wr) / 1024))
PY
