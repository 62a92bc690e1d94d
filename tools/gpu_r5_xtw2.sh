#!/bin/bash
# Round 5: fusedxtw420_kernel asked for 256 registers (two workgroups per CU, spills to scratch) against the 420-register build
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/xtw2"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for lib in "" "$ROOT/tools/ab/libmijpeg_xtw2.so" "" "$ROOT/tools/ab/libmijpeg_xtw2.so"; do
  tag=base; [ -n "$lib" ] && tag=xtw2
  MIJPEG_LIBRARY="$lib" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$tag" -o t -- python $ROOT/tools/xt_launches.py --hidden --launches 40 > "$OUT/trace_$tag.log" 2>&1
  echo "$tag: $(grep fusedxtw $OUT/trace_$tag/t_kernel_stats.csv | cut -d, -f1-7 | tail -1)"
done
cd "$ROOT"
MIJPEG_LIBRARY="$ROOT/tools/ab/libmijpeg_xtw2.so" timeout 600 python -m pytest tests -m gpu -q -k "xt" -x 2>&1 | tail -2
