#!/bin/bash
# Round 5: the 12-bit 4:4:4 / 4:2:2 kernels' 48-byte line pieces with and without the non-temporal hint
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5v; export TMPDIR=/tmp
O=gpurun_out/r5v
for round in 1 2 3; do
for v in "" f12t; do
  echo "== variant: ${v:-product (non-temporal)}" >> $O/f12_stores.txt
  if [ -n "$v" ]; then export MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_$v.so; else unset MIJPEG_LIBRARY; fi
  LAYOUTS=444_12,422_12 timeout 600 python tools/layout_bench.py 2>&1 | grep -v amdgpu.ids >> $O/f12_stores.txt
done
done
cat $O/f12_stores.txt
