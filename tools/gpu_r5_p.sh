#!/bin/bash
# Round 5: per-frame tables through the 12-bit kernels, alpha tests, then the random campaign over every kernel family
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5p; export TMPDIR=/tmp
O=gpurun_out/r5p
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_xt_alpha.py -m gpu -q -x -k "per_frame_tables or alpha" > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest_gpu.log
N=1600 SEED=20260930 timeout 1200 python tools/random_campaign.py > $O/random_campaign.txt 2>&1; echo "campaign exit $?"; grep -v "Suspension\|amdgpu.ids" $O/random_campaign.txt | tail -12
