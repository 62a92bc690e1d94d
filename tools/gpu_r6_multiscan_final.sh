#!/bin/bash
# Round 6, end: the device decoder of progressive frames / hidden refinement scans after the second pass over it (streams from build/ms,
# tools/multiscan_probe.py make): read times, the host-side steps of one read, the kernels under rocprofv3 --kernel-trace --stats
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/msfinal; export TMPDIR=/tmp
O=$ROOT/gpurun_out/msfinal
for i in 1 2 3; do timeout 200 python tools/multiscan_probe.py run build/ms 8 2>&1 | grep "ran on"; done | tee $O/reads.txt
MIJPEG_READ_TIMES=1 MIJPEG_TRACE_SUBMIT=1 timeout 200 python tools/multiscan_probe.py run build/ms 4 2>&1 | grep -v amdgpu.ids > $O/steps.txt
for S in xt4k_rR4_z8 prog8k_z8; do
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$S -o ms -- python "$ROOT/tools/multiscan_probe.py" run "$ROOT/build/ms" 4 $S > $O/$S.log 2>&1 )
  f=$(find $O/$S -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { echo $S; cat "$f" | head -8; }
  t=$(find $O/$S -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python - "$t" <<'P'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "huffman_prog" in r["Kernel_Name"] or "coef_range" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // 8
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print(f"  {r['Kernel_Name'].split('::')[1].split('(')[0]:34s} start {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} us  takes {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us  grid {r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size', '?')}")
P
done 2>&1 | tee $O/kernels.txt
