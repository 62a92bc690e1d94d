#!/bin/bash
# rocprofv3 kernel statistics of the encoder direction (forward kernels + device entropy coder)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/prof_encoder"
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
LAYOUTS=420 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t" -o t -- python $ROOT/tools/forward_bench.py > "$OUT/log.txt" 2>&1
grep -v amdgpu.ids "$OUT/log.txt" | grep "encode one\|Gpixel\|resident in HBM" | head -12
python - <<PY
import csv, glob
f = glob.glob("$OUT/t/**/t_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
print("   kernel, calls, average us, total ms")
for r in rows:
    if "rocclr" in r["Name"]: continue
    print("   %-70s %6s %10.1f %10.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
