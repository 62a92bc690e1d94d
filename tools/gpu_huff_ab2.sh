#!/bin/bash
# huffman_scan_kernel alone (one decoder object, 32 x 4K frames per launch): library as built against a variant (MIJPEG_LIBRARY)
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export CFG_FRAMES=64 SETTINGS=32x1 STEPS=3
for lib in "" "$R/tools/exp/$1"; do
  export MIJPEG_LIBRARY=$lib
  rm -rf /tmp/ab2
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab2 -o t -- python $R/tools/batch4k_bench.py > /tmp/ab2.log 2>&1
  echo "lib=$lib: $(find /tmp/ab2 -name '*kernel_stats.csv' -exec grep huffman_scan {} \; | cut -d, -f2-7)"
  grep "per batch" /tmp/ab2.log | head -2
done
cd $R; MIJPEG_LIBRARY=$R/tools/exp/$1 timeout 300 python -m pytest tests -m gpu -x -q -k "entropy or huff or batch4k or device or restart" 2>&1 | tail -3
