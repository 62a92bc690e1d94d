#!/bin/bash
# kernel duration of huffman_scan_kernel per lanes-per-wave setting (tuning aid)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp; export TMPDIR=/tmp
for L in ${LANES:-1 2 4 8 16 32 64}; do
  MIJPEG_HUFF_LANES=$L timeout 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/prof_huffL$L -o t -- python $ROOT/tools/entropy_bench.py > /dev/null 2>&1
  python - <<PY
import csv
r=[x for x in csv.DictReader(open("$ROOT/gpurun_out/prof_huffL$L/t_kernel_trace.csv")) if "huffman" in x["Kernel_Name"]]
d=[int(x["End_Timestamp"])-int(x["Start_Timestamp"]) for x in r]
rep=int("${MIJPEG_HUFF_REPEAT:-1}")
print("L=$L", [round(min(d[i:i+6*rep])/1e3) for i in range(0, len(d), 6*rep)], "us per RI in ${RI} (min over launches)")
PY
done
