#!/bin/bash
# Round 5: are config 4's sporadic slow steps periods in which the container's CPU quota ran out?
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5q3
{ echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max)"; CFG_FRAMES=256 SETTINGS=24x4 STEPS=60 timeout 600 python tools/batch_stall_probe.py 2>&1 | grep -v amdgpu.ids;
  echo "--- MIJPEG_THREADS=16"; MIJPEG_THREADS=16 CFG_FRAMES=256 SETTINGS=24x4 STEPS=60 timeout 600 python tools/batch_stall_probe.py 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r5q3/batch_quota.txt
cut -c1-1200 gpurun_out/r5q3/batch_quota.txt | head -40
