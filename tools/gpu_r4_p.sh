#!/bin/bash
# Round 4, visit p: request sequences on JPEG XT frames (two request models, row maps for six planes), whole GPU suite
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r4p; export TMPDIR=/tmp
O=gpurun_out/r4p
timeout 600 python -m pytest tests/test_rect_calls_xt.py -m gpu -q -x > $O/pytest_xt_rect.log 2>&1; echo "xt rect exit $?"; tail -15 $O/pytest_xt_rect.log
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -8 $O/pytest_gpu.log
