#!/bin/bash
# Round 5: the hint on the 3 x 8-byte line pieces of fused444_kernel and the unpacked fused420_kernel
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5x; export TMPDIR=/tmp
O=gpurun_out/r5x
for round in 1 2 3; do
for v in "" t444; do
  echo "== variant: ${v:-product}" >> $O/nt_stores_8bit.txt
  if [ -n "$v" ]; then export MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_$v.so; else unset MIJPEG_LIBRARY; fi
  LAYOUTS=444,420_12 timeout 600 python tools/layout_bench.py 2>&1 | grep "4.*:" | cut -c1-170 >> $O/nt_stores_8bit.txt
  SATURATED=1 LAYOUTS=420 timeout 600 python tools/layout_bench.py 2>&1 | grep "420:" | cut -c1-170 >> $O/nt_stores_8bit.txt
done
done
cat $O/nt_stores_8bit.txt
