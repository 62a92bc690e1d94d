#!/bin/bash
# Round 4, visit k: run length of the tile order in the real kernel, the JPEG XT kernels with both orders, 12-bit 4:2:0 back on the old one
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r4k; export TMPDIR=/tmp
O=gpurun_out/r4k
REPS=3 bash tools/gpu_hl_variants.sh r4k row run30 run15 run8
for rep in 1 2; do for v in row xt2; do echo "xt $v $rep: $(MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_$v.so timeout 300 python tools/xt_bench.py 2>&1 | grep -E 'ms/launch|bit-exact' | cut -c1-120 | tr '\n' ' ')"; done; done | tee $O/xt_order.txt
LAYOUTS=420_12,444 MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_row.so timeout 300 python tools/layout_bench.py 2>&1 | grep "ms/launch" | tee $O/layouts.txt | cut -c1-150
