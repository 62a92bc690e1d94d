#!/bin/bash
# kernel + memcpy trace of the config 4 batch path -> gpurun_out/prof_batch4k/
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/prof_batch4k"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 240 python $ROOT/tools/batch4k_bench.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/plain.txt"
SETTINGS=16x4 STEPS=2 timeout 240 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d "$OUT/trace" -o t -- python $ROOT/tools/batch4k_bench.py > "$OUT/trace.log" 2>&1; echo "trace exit $?"
grep -v amdgpu.ids "$OUT/trace.log" | tail -3
for f in $(find "$OUT/trace" -name "*_stats.csv"); do echo "== $f"; head -12 "$f"; done
