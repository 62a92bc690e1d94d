#!/bin/bash
# Round 5: where fused_tile_kernel's time goes -- phase A alone, phase B alone, both
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5q; export TMPDIR=/tmp
O=gpurun_out/r5q
for v in "" skipa skipb; do
  echo "== variant: ${v:-product}" >> $O/tile_phases.txt
  if [ -n "$v" ]; then export MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_$v.so; else unset MIJPEG_LIBRARY; fi
  LAYOUTS=3x1,1x4,3x3,lumasub timeout 600 python tools/layout_bench.py 2>&1 | grep -v amdgpu.ids >> $O/tile_phases.txt
done
cat $O/tile_phases.txt
