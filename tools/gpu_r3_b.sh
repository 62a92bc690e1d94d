#!/bin/bash
# Round 3, visit B: two-level Huffman tables + upload decoupled from the reconstruction kernel + ramped chunk schedule.
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r3b; export TMPDIR=/tmp
echo "== pytest gpu (entropy / batch / damaged)"; timeout 900 python -m pytest tests -m gpu -x -q -k "entropy or huff or batch or walk or damage or device or restart" > gpurun_out/r3b/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/r3b/pytest_gpu.log
echo "== batch4k plain"; ( cd /tmp; CFG_FRAMES=256 SETTINGS=24x4,24x4r,32x4r,32x3r,16x4r,24x3r STEPS=4 timeout 300 python $ROOT/tools/batch4k_bench.py 2>&1 | grep -v amdgpu.ids | tee $ROOT/gpurun_out/r3b/batch4k_plain.txt )
echo "== batch4k trace"; ( cd /tmp; CFG_FRAMES=256 SETTINGS=24x4r STEPS=3 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $ROOT/gpurun_out/r3b/trace -o t -- python $ROOT/tools/batch4k_bench.py > $ROOT/gpurun_out/r3b/trace.log 2>&1; echo "trace exit $?" )
head -6 gpurun_out/r3b/trace/t_kernel_stats.csv
echo "== entropy bench"; ( cd /tmp; timeout 300 python $ROOT/tools/entropy_bench.py 2>&1 | grep -v amdgpu.ids | tail -25 | tee $ROOT/gpurun_out/r3b/entropy_bench.txt )
