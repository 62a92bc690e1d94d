#!/bin/bash
# Round 5: speculative batch reconstruction -- tests, then config 4 step by step with and without it
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5h; export TMPDIR=/tmp
O=gpurun_out/r5h
timeout 900 python -m pytest tests/test_batch4k.py tests/test_sharding.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -15 $O/pytest.log
for spec in on off; do
  if [ $spec = off ]; then export MIJPEG_BATCH_NO_SPECULATION=1; else unset MIJPEG_BATCH_NO_SPECULATION; fi
  echo "== speculation $spec"
  CFG_FRAMES=256 SETTINGS=24x4,24x4r,28x4r,32x4,16x2,16x4,8x4r STEPS=12 timeout 600 python tools/batch_stall_probe.py 2>&1 | grep -v "^   step\|slowest\|amdgpu.ids\|shader clock" | cut -c1-200 | tee $O/probe256_$spec.txt
  CFG_FRAMES=32 SETTINGS=8x4,11x3,16x2 STEPS=12 timeout 300 python tools/batch_stall_probe.py 2>&1 | grep -v "^   step\|slowest\|amdgpu.ids\|shader clock" | cut -c1-200 | tee $O/probe32_$spec.txt
done
