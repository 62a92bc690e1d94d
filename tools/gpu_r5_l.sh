#!/bin/bash
# Round 5: the whole GPU suite after the box-campaign fixes
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5l; export TMPDIR=/tmp
O=gpurun_out/r5l
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -15 $O/pytest_gpu.log
