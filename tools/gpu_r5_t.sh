#!/bin/bash
# Round 5: damaged L-only / alpha / lossless / grey / integer XT streams, product pixels on the device against the oracle's
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5t; export TMPDIR=/tmp
for s in 3 4 5; do SEED=$s PER_FILE=14 timeout 1200 python tools/xt_gpu_damage_campaign.py 2>&1 | grep -v "amdgpu.ids\|Suspension"; done | tee gpurun_out/r5t/xt_gpu_damage_campaign.txt | tail -40
