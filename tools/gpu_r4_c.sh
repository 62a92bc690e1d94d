#!/bin/bash
# Round 4, visit c: GPU suite with the shifted store path and the no-residual merge, W = 7678 layouts, random campaign
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r4c; export TMPDIR=/tmp
O=gpurun_out/r4c
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest_gpu.log
echo "== layouts W=7678"; for W in 7680 7678 7677 7679; do W=$W LAYOUTS=420 timeout 300 python tools/layout_bench.py 2>&1 | grep "ms/launch" | sed "s/^/W=$W /"; done | tee $O/layouts_w.txt | cut -c1-170
echo "== campaign"; timeout 900 python tools/random_campaign.py > $O/random_campaign.txt 2>&1; tail -12 $O/random_campaign.txt
