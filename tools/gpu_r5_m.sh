#!/bin/bash
# Round 5: 12-bit 4:4:4 on a fused kernel of its own (fused444_12_kernel): parity, then rates beside the tile kernel it replaces
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5m; export TMPDIR=/tmp
O=gpurun_out/r5m
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "12bit or 12_bit or fused_tile or random_layout or dnl" > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest_gpu.log
LAYOUTS=444_12,422_12,420_12,444,422 timeout 600 python tools/layout_bench.py > $O/layouts12.txt 2>&1
MIJPEG_NO_F444_12=1 MIJPEG_NO_F422_12=1 LAYOUTS=444_12,422_12 timeout 600 python tools/layout_bench.py >> $O/layouts12.txt 2>&1
cat $O/layouts12.txt
