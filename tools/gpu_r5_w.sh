#!/bin/bash
# Round 5: damaged progressive and hidden-bit streams through the host pipeline windows (masks, appliers, early rows) on the device
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5w
for s in 11 12; do CLASSES=refinement PER_FILE=40 SEED=$s timeout 600 python tools/xt_gpu_damage_campaign.py 2>&1 | grep -v amdgpu.ids | tail -6; done > gpurun_out/r5w/refinement_damage.txt
cat gpurun_out/r5w/refinement_damage.txt
