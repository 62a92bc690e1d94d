#!/bin/bash
# Round 5: config 4 step by step under the profiler: what is a slow step slow in (uploads? kernels?)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5g; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r5g
cd /tmp
for S in 16x2 24x4r 8x4r; do
  CFG_FRAMES=256 SETTINGS=$S STEPS=14 WARMUP=2 IDLE_MS=6 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr_$S -o t -- python $ROOT/tools/batch_stall_probe.py > $O/probe_$S.log 2>&1
  grep "steps ms" $O/probe_$S.log | cut -c1-220
  python $ROOT/tools/batch_step_trace.py $O/tr_$S > $O/steps_$S.txt 2>&1; cat $O/steps_$S.txt
  rm -rf $O/tr_$S
done
