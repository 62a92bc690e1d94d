#!/bin/bash
# Round 5: config 4 -- catch (a) a hiccup step (one step of +6..8 ms) with its per-chunk submit times, (b) the slow first steps of 16 x 2 in a trace
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5j; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r5j
CFG_FRAMES=256 SETTINGS=24x4,16x4 STEPS=60 timeout 600 python tools/batch_stall_probe.py 2>&1 | grep -v "amdgpu.ids\|shader clock" | cut -c1-330 > $O/hiccups.txt; cat $O/hiccups.txt | cut -c1-250
cd /tmp
CFG_FRAMES=256 SETTINGS=16x2 STEPS=14 WARMUP=0 IDLE_MS=6 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/tr -o t -- python $ROOT/tools/batch_stall_probe.py > $O/probe_16x2.log 2>&1
grep "steps ms" $O/probe_16x2.log | cut -c1-220
python $ROOT/tools/batch_step_trace.py $O/tr > $O/steps_16x2.txt 2>&1; cat $O/steps_16x2.txt
python $ROOT/tools/batch_timeline.py $O/tr 16 > $O/last_batch_16x2.txt 2>&1
rm -rf $O/tr
