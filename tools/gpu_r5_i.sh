#!/bin/bash
# Round 5: (1) the pipeline's wait polls before it blocks: config 4 step by step, with / without; (2) two AC symbols per table step in
# the walk kernels: tests of the streams without restart markers, then the single-frame and batch timings with / without
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5i; export TMPDIR=/tmp
O=gpurun_out/r5i
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_batch4k.py -m gpu -q -x -k "walk or restart or nodri or no_dri or virtual or device_entropy or batch or speculat or submit" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -5 $O/pytest.log
for spin in on off; do
  if [ $spin = off ]; then export MIJPEG_NO_SPIN_WAIT=1; else unset MIJPEG_NO_SPIN_WAIT; fi
  echo "== polling wait $spin"
  CFG_FRAMES=256 SETTINGS=24x4,28x4r,32x4,16x2,16x4 STEPS=14 timeout 600 python tools/batch_stall_probe.py 2>&1 | grep -v "^   step\|slowest\|amdgpu.ids\|shader clock" | cut -c1-220 | tee $O/probe256_spin_$spin.txt
done
unset MIJPEG_NO_SPIN_WAIT
for pairs in on off; do
  if [ $pairs = off ]; then export MIJPEG_WALK_NO_PAIRS=1; else unset MIJPEG_WALK_NO_PAIRS; fi
  echo "== walk pairs $pairs"
  timeout 300 python tools/nodri_timeline.py 2>&1 | grep -v amdgpu.ids | head -4 | tee $O/nodri_pairs_$pairs.txt
  FRAMES=16 timeout 300 python tools/nodri_timeline.py 2>&1 | grep -v amdgpu.ids | head -4 | tee -a $O/nodri_pairs_$pairs.txt
done
