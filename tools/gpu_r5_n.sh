#!/bin/bash
# Round 5: the progressive residual type and residual refinement scans on the GPU + the XT files again
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5n; export TMPDIR=/tmp
O=gpurun_out/r5n
timeout 1200 python -m pytest tests/test_xt_lossless.py tests/test_xt_alpha.py tests/test_xt_boxes.py tests/test_xt_lonly.py tests/test_xt_damaged.py tests/test_xt_general.py tests/test_xt_int8.py -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -25 $O/pytest_gpu.log
