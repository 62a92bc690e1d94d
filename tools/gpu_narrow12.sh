#!/bin/bash
# Round 6, last session: the 12-bit kernels' colour stage in one 32-bit sum per channel (colour12<true>, narrow12_colour).
# Parity of the 12-bit tests, then an alternating A-B against the two-step flavour (MIJPEG_NO_NARROW12=1). -> gpurun_out/narrow12
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/narrow12; export TMPDIR=/tmp
O=gpurun_out/narrow12
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "12bit or 12 or per_frame_tables" > $O/pytest_12bit.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_12bit.log
for round in 1 2; do
  for off in 0 1; do
    if [ $off = 1 ]; then export MIJPEG_NO_NARROW12=1; else unset MIJPEG_NO_NARROW12; fi
    echo "== round $round MIJPEG_NO_NARROW12=${MIJPEG_NO_NARROW12:-}" >> $O/ab.txt
    LAYOUTS=420_12,444_12,422_12 timeout 600 python tools/layout_bench.py 2>&1 | grep -v amdgpu.ids >> $O/ab.txt
  done
done
unset MIJPEG_NO_NARROW12
cat $O/ab.txt
