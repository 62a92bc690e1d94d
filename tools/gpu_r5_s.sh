#!/bin/bash
# Round 5: encoder-written JPEG XT files with random switches on the device against the oracle
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5s; export TMPDIR=/tmp
N=4000 SEED=20261002 timeout 1500 python tools/xt_gpu_campaign.py 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r5s/xt_gpu_campaign.txt | tail -40
