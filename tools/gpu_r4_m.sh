#!/bin/bash
# Round 4, visit m: config 4 (256 x 4K streams -> pixels in HBM) with the kernel variants of this round, alternating on one box
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r4m; export TMPDIR=/tmp
O=gpurun_out/r4m
for rep in 1 2; do for v in cur base o1 w3; do
  echo "== $v $rep"; MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_$v.so CFG_FRAMES=256 STEPS=5 SETTINGS=24x4,32x4 timeout 600 python tools/batch4k_bench.py 2>&1 | grep "ms per batch" | cut -c1-110
done; done | tee $O/batch4k_variants.txt
