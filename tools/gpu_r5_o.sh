#!/bin/bash
# Round 5: L-only files whose height arrives in a DNL marker
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5o; export TMPDIR=/tmp
O=gpurun_out/r5o
timeout 1200 python -m pytest tests/test_xt_lonly.py tests/test_dnl.py -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -40 $O/pytest_gpu.log
