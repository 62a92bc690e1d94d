#!/bin/bash
# huffman_scan_kernel alone (one decoder object, 32 x 4K frames per launch) + the no-DRI single 8K frame
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export CFG_FRAMES=64 SETTINGS=32x1 STEPS=3
rm -rf /tmp/ht
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ht -o t -- python $R/tools/batch4k_bench.py > /tmp/ht.log 2>&1
echo "32 x 4K: $(find /tmp/ht -name '*kernel_stats.csv' -exec grep huffman_scan {} \; | cut -d, -f1-4,6-7)"
python $R/tools/nodri_timeline.py 2>&1 | tail -1
FRAMES=16 python $R/tools/nodri_timeline.py 2>&1 | tail -1
