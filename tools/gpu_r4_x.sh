#!/bin/bash
# Round 4, visit x: luma prefetch in the unpacked 4:2:0 kernel (chroma beyond the 16-bit gate, 12-bit frames): parity, then A-B
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r4x; export TMPDIR=/tmp
O=gpurun_out/r4x
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "420 or golden or extreme or adversarial or 12bit or xt" 2>&1 | tail -2
REPS=3 bash tools/gpu_hl_variants.sh r4x nopf cur
for rep in 1 2 3; do for v in nopf cur; do echo "$v $rep $(LAYOUTS=420_12 MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_$v.so timeout 300 python tools/layout_bench.py 2>&1 | grep 'ms/launch' | cut -c1-10,60-130)"; done; done | tee $O/layout_420_12.txt
