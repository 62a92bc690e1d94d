#!/bin/bash
# Round 4, visit o: luma prefetch in the 4:2:2 / 4:4:0 kernels (parity, then both layouts with and without), small launches of the headline kernel
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r4o; export TMPDIR=/tmp
O=gpurun_out/r4o
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "422 or 440 or golden or full_size or random" 2>&1 | tail -2
for rep in 1 2 3; do for v in nopf cur; do
  if [ $v = nopf ]; then export MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_nopf.so; else unset MIJPEG_LIBRARY; fi
  LAYOUTS=422,440 timeout 300 python tools/layout_bench.py 2>&1 | grep 'ms/launch' | cut -c1-8,60-140 | sed "s/^/$v $rep /"
done; done | tee $O/prefetch_422_440.txt
unset MIJPEG_LIBRARY
for fr in 1 8 64; do
  timeout 300 python bench.py --frames $fr --workload headline --no-cpu-baseline --no-end-to-end --no-traffic --no-xt --no-dense --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('frames', $fr, 'frac', r['roofline']['frac'], 'kernel_ms', r['roofline']['kernel_ms'])"
done | tee $O/small_launches.txt
