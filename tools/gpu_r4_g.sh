#!/bin/bash
# Round 4, visit g: the traffic-only sweep (tile shapes, tile order, directions, store hints), temporal stores in the real kernel
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r4g; export TMPDIR=/tmp
timeout 300 tools/microbench/stream_ceiling | tee gpurun_out/r4g/stream_ceiling.txt
REPS=3 bash tools/gpu_hl_variants.sh r4g pf0 pf2w4 pf2w4t
