#!/bin/bash
# Round 5: is config 4's stall mode the clock governor?  Long warm-up, idle gaps, shader clock beside every step
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5f; export TMPDIR=/tmp
O=gpurun_out/r5f
ls /sys/class/drm/card*/device/pp_dpm_sclk 2>&1 | head -3; cat /sys/class/drm/card*/device/pp_dpm_sclk 2>&1 | head -12
echo "== warm-up 2 (as the bench did)";  CFG_FRAMES=256 SETTINGS=16x2,24x4 STEPS=12 WARMUP=2 timeout 300 python tools/batch_stall_probe.py 2>&1 | grep -v "^   step\|slowest\|amdgpu.ids" | tee $O/warm2.txt
echo "== warm-up 25";                     CFG_FRAMES=256 SETTINGS=16x2,24x4 STEPS=12 WARMUP=25 timeout 300 python tools/batch_stall_probe.py 2>&1 | grep -v "^   step\|slowest\|amdgpu.ids" | tee $O/warm25.txt
echo "== warm-up 25, 300 ms idle before every step"; CFG_FRAMES=256 SETTINGS=16x2,24x4 STEPS=8 WARMUP=25 IDLE_MS=300 timeout 300 python tools/batch_stall_probe.py 2>&1 | grep -v "^   step\|slowest\|amdgpu.ids" | tee $O/idle300.txt
