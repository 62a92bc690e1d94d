#!/bin/bash
# huffman_scan_kernel alone (one decoder object, 32 x 4K frames per launch): two-level tables against the canonical walk
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/huff_ab
export CFG_FRAMES=64 SETTINGS=32x1 STEPS=3
for v in sub nosub; do
  if [ $v = nosub ]; then export MIJPEG_HUFF_NO_SUBTABLES=1; else unset MIJPEG_HUFF_NO_SUBTABLES; fi
  rm -rf /tmp/ab_$v
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -o t -- python $R/tools/batch4k_bench.py > /tmp/ab_$v.log 2>&1
  echo "$v: $(find /tmp/ab_$v -name '*kernel_stats.csv' -exec grep huffman_scan {} \; | cut -d, -f2-7)"
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/abp_$v -o t -- python $R/tools/batch4k_bench.py > /tmp/abp_$v.log 2>&1
  python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/abp_$v/**/t_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "huffman_scan" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$v", {k: round(sum(x) / len(x)) for k, x in sorted(acc.items())})
PY
done 2>&1 | tee $R/gpurun_out/huff_ab/result.txt
