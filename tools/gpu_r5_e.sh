#!/bin/bash
# Round 5: config 4's stall mode, step by step (tools/batch_stall_probe.py)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5e; export TMPDIR=/tmp
O=gpurun_out/r5e
CFG_FRAMES=256 SETTINGS=24x4,24x4r,28x4r,32x4,16x2,8x4r STEPS=12 timeout 600 python tools/batch_stall_probe.py > $O/probe256.txt 2>&1; tail -60 $O/probe256.txt
CFG_FRAMES=32 SETTINGS=8x4,11x3,16x3,16x2 STEPS=12 timeout 300 python tools/batch_stall_probe.py > $O/probe32.txt 2>&1; tail -40 $O/probe32.txt
