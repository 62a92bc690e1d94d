#!/bin/bash
# Round 4, visit aa: rocprofv3 kernel + memcpy timeline of config 4 (256 x 4K): pixels left in HBM, and pixels into pinned host memory (full duplex)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r4aa; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r4aa
cd /tmp
CFG_FRAMES=256 SETTINGS=24x4 STEPS=3 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace_hbm -o t -- python $ROOT/tools/batch4k_bench.py > $O/trace_hbm.log 2>&1; echo "trace hbm exit $?"; grep "ms per batch" $O/trace_hbm.log | cut -c1-100
python $ROOT/tools/batch_timeline.py $O/trace_hbm 11 > $O/batch4k_timeline.txt 2>&1; tail -4 $O/batch4k_timeline.txt
DOWNLOAD=1 CFG_FRAMES=256 SETTINGS=32x4 STEPS=2 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace_host -o t -- python $ROOT/tools/batch4k_bench.py > $O/trace_host.log 2>&1; echo "trace host exit $?"; grep "pinned host" $O/trace_host.log | cut -c1-140
python $ROOT/tools/batch_timeline.py $O/trace_host 8 > $O/batch4k_full_duplex_timeline.txt 2>&1; tail -4 $O/batch4k_full_duplex_timeline.txt
rm -rf $O/trace_hbm $O/trace_host
