#!/bin/bash
# Round 4, visit f: what HBM gives the headline launch's traffic without the arithmetic (tools/microbench/stream_ceiling), then the
# luma prefetch + four workgroups per CU build with and without the rotated LDS lines against the build without either
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r4f; export TMPDIR=/tmp
timeout 120 tools/microbench/stream_ceiling | tee gpurun_out/r4f/stream_ceiling.txt
REPS=3 bash tools/gpu_hl_variants.sh r4f pf0 pf2w4 rotpf2w4
timeout 120 tools/microbench/stream_ceiling | tee -a gpurun_out/r4f/stream_ceiling.txt
