#!/bin/bash
# A-B: fused420_kernel<12> with the luma blocks requested in front of phase A (tools/ab/libmijpeg_pf12.so: -DF420_12_PREFETCH=1; pf12w3: and F420_12_MINW=3)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/pf12; export TMPDIR=/tmp
O=gpurun_out/pf12/ab.txt; : > $O
for round in 1 2 3; do
  for lib in "" tools/ab/libmijpeg_pf12.so tools/ab/libmijpeg_pf12w3.so; do
    [ -n "$lib" ] && [ ! -f "$lib" ] && continue
    if [ -n "$lib" ]; then export MIJPEG_LIBRARY=$ROOT/$lib; else unset MIJPEG_LIBRARY; fi
    echo "== round $round library ${lib:-default}" >> $O
    LAYOUTS=420_12 timeout 600 python tools/layout_bench.py 2>&1 | grep -v amdgpu.ids >> $O
  done
done
cat $O
