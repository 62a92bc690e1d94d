#!/bin/bash
# Round 5, first visit: the whole GPU suite with the L-only files (tests/test_xt_lonly.py, tests/test_spec_boxes.py)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5a; export TMPDIR=/tmp
O=gpurun_out/r5a
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -25 $O/pytest_gpu.log
