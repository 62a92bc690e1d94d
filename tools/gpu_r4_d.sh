#!/bin/bash
# Round 4, visit d: GPU suite with the 8-bit integer JPEG XT files and the truncated-XT fill, W = 767x layouts with temporal
# stores on misaligned lines
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r4d; export TMPDIR=/tmp
O=gpurun_out/r4d
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -15 $O/pytest_gpu.log
echo "== layouts W=767x"; for W in 7680 7678 7677 7679; do W=$W LAYOUTS=420 timeout 300 python tools/layout_bench.py 2>&1 | grep "ms/launch" | sed "s/^/W=$W /"; done | tee $O/layouts_w.txt | cut -c1-170
