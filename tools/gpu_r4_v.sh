#!/bin/bash
# Round 4, visit v: the whole GPU suite and the random differential campaign on the final kernels
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r4v; export TMPDIR=/tmp
O=gpurun_out/r4v
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest_gpu.log
timeout 900 python tools/random_campaign.py > $O/random_campaign.txt 2>&1; echo "campaign exit $?"; tail -12 $O/random_campaign.txt
