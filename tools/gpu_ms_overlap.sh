#!/bin/bash
# Round 6: device_entropy_multiscan with the uploads level by level on the copy stream (the first launches run while the rest of the
# file is gathered and brought up) against the build before it (tools/ab/libmijpeg_oldms.so); streams from build/ms (multiscan_probe.py make)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/msov; export TMPDIR=/tmp
O=gpurun_out/msov
timeout 900 python -m pytest tests/test_device_multiscan.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest.log
for i in 1 2 3; do
  echo "--- new"; timeout 200 python tools/multiscan_probe.py run build/ms 8 2>&1 | grep "prefer-gpu"
  echo "--- old"; MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_oldms.so timeout 200 python tools/multiscan_probe.py run build/ms 8 2>&1 | grep "prefer-gpu"
done | tee $O/ab.txt
MIJPEG_TRACE_SUBMIT=1 timeout 200 python tools/multiscan_probe.py run build/ms 3 xt4k_rR4_z8 > $O/trace.txt 2>&1; tail -40 $O/trace.txt
