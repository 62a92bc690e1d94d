#!/bin/bash
# Round 4, visit i: the whole GPU suite with the new tile order (all fused kernels) and the luma prefetch, every layout's kernel
# with the old order (tools/ab/libmijpeg_ord1.so) and the new one, the headline launch A-B
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r4i; export TMPDIR=/tmp
O=gpurun_out/r4i
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest_gpu.log
echo "== layouts, old order"; MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_ord1.so timeout 600 python tools/layout_bench.py 2>&1 | grep "ms/launch" | tee $O/layouts_old_order.txt | cut -c1-150
echo "== layouts, new order"; timeout 600 python tools/layout_bench.py 2>&1 | grep "ms/launch" | tee $O/layouts_new_order.txt | cut -c1-150
cp libjpeg_amd/libmijpeg.so tools/ab/libmijpeg_new.so
REPS=3 bash tools/gpu_hl_variants.sh r4i ord1 new
