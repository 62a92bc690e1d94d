#!/bin/bash
# Round 5: rocprofv3 evidence for the two new 12-bit kernels (kernel trace + stats, FETCH_SIZE / WRITE_SIZE / SQ passes; separate --pmc passes)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/r5u"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/tools/layout_bench.py"
export LAYOUTS=444_12,422_12
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- $CMD > "$OUT/trace.log" 2>&1; echo "trace exit $?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o t -- $CMD > "$OUT/fetch.log" 2>&1; echo "fetch exit $?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o t -- $CMD > "$OUT/write.log" 2>&1; echo "write exit $?"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc_sq" -o t -- $CMD > "$OUT/sq.log" 2>&1; echo "sq exit $?"
cd "$ROOT"
grep -h "fused4" $OUT/trace/*kernel_stats.csv
python tools/summarize_prof.py "$OUT" 2>/dev/null | grep -A6 "fused4" | head -60
