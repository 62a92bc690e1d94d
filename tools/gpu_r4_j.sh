#!/bin/bash
# Round 4, visit j: GPU suite again (XT launch grid fixed), run-length sweep of the traffic-only kernel, the remaining layouts with both orders
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r4j; export TMPDIR=/tmp
O=gpurun_out/r4j
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest_gpu.log
timeout 300 tools/microbench/stream_ceiling | tee $O/stream_ceiling.txt | head -40
L="411,gray,cmyk,3x1,lumasub,3x3,420_12,444_12"
echo "== layouts, old order"; LAYOUTS=$L MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_ord1.so timeout 600 python tools/layout_bench.py 2>&1 | grep "ms/launch" | tee $O/layouts_old_order.txt | cut -c1-150
echo "== layouts, new order"; LAYOUTS=$L timeout 600 python tools/layout_bench.py 2>&1 | grep "ms/launch" | tee $O/layouts_new_order.txt | cut -c1-150
