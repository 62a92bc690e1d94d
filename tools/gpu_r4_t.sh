#!/bin/bash
# Round 4, visit t: fused_flat_kernel (CMYK, RGB stored as such): parity, then the layouts with the tile kernel / two and three workgroups per CU
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r4t; export TMPDIR=/tmp
O=gpurun_out/r4t
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "flat or fused_tile or cmyk or golden" 2>&1 | tail -4
for rep in 1 2; do
  echo "tile  $rep $(MIJPEG_NO_FUSED_FLAT=1 LAYOUTS=cmyk timeout 300 python tools/layout_bench.py 2>&1 | grep 'ms/launch' | cut -c1-170)"
  echo "flat2 $rep $(LAYOUTS=cmyk timeout 300 python tools/layout_bench.py 2>&1 | grep 'ms/launch' | cut -c1-170)"
  echo "flat3 $rep $(MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_flat3.so LAYOUTS=cmyk timeout 300 python tools/layout_bench.py 2>&1 | grep 'ms/launch' | cut -c1-170)"
  echo "rgb tile $rep $(MIJPEG_NO_FUSED_FLAT=1 FLAGS=1 LAYOUTS=444 timeout 300 python tools/layout_bench.py 2>&1 | grep 'ms/launch' | cut -c1-170)"
  echo "rgb flat $rep $(FLAGS=1 LAYOUTS=444 timeout 300 python tools/layout_bench.py 2>&1 | grep 'ms/launch' | cut -c1-170)"
done | tee $O/flat_layouts.txt
