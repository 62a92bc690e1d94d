#!/bin/bash
# huffman_prog_kernel under rocprofv3 --kernel-trace --stats for several lanes-per-wave settings (MIJPEG_HUFF_LANES), config 5 -rR 4 -z 8
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/proglanes; export TMPDIR=/tmp
O=$ROOT/gpurun_out/proglanes
for L in ${LANES:-8 16 32 64}; do
  ( cd /tmp && MIJPEG_HUFF_LANES=$L timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/l$L -o ms -- python "$ROOT/tools/multiscan_probe.py" run "$ROOT/build/ms" 4 ${STREAMS:-xt4k_rR4_z8} > $O/l$L.log 2>&1 )
  f=$(find $O/l$L -name "*kernel_stats.csv" | head -1)
  echo "lanes $L"; grep "prefer-gpu" $O/l$L.log; [ -n "$f" ] && grep -i "huffman_prog\|coef_range" "$f"
  t=$(find $O/l$L -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python - "$t" <<'P'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "huffman_prog" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last read's launches
n = len(rows) // 8 if len(rows) >= 8 else len(rows)
last = rows[-n:] if n else rows
print("  last read:", " ".join(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.0f}" for r in last), "us; grid", " ".join(r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?") for r in last))
P
done
