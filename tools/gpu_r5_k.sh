#!/bin/bash
# Round 5: lossless JPEG XT on the GPU + every XT test file again (the merge kernels and the specification parser changed)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5k; export TMPDIR=/tmp
O=gpurun_out/r5k
timeout 1200 python -m pytest tests/test_xt_lossless.py tests/test_xt_alpha.py tests/test_xt_boxes.py tests/test_xt_lonly.py tests/test_xt_damaged.py tests/test_xt_general.py tests/test_xt_grey.py tests/test_xt_int8.py tests/test_xt_noct.py tests/test_spec_boxes.py tests/test_rect_calls_xt.py -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -25 $O/pytest_gpu.log
