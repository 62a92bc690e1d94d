#!/bin/bash
# Round 4, visit n: tile order at other frame sizes (4K, 1080p, 8K; 8 frames per launch), parity of the magic-number tile position
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r4n; export TMPDIR=/tmp
O=gpurun_out/r4n
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_batch4k.py tests/test_xt_int8.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2 3; do for v in cur o1; do for wh in "3840 2160" "1920 1080" "7680 4320"; do set -- $wh
  echo "$v $rep ${1}x$2: $(W=$1 H=$2 LAYOUTS=420 MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_$v.so timeout 300 python tools/layout_bench.py 2>&1 | grep 'ms/launch' | cut -c60-130)"
done; done; done | tee $O/order_by_size.txt
REPS=2 bash tools/gpu_hl_variants.sh r4n cur o1
