#!/bin/bash
# huffman_scan_kernel on ONE 8K 4:2:0 DRI=8 frame (8100 restart intervals): kernel duration per lanes-per-wave setting
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cat > /tmp/one8k.py <<PY
import sys, time
sys.path.insert(0, "$R")
from libjpeg_amd import api, synth
data = synth.synth_jpeg(7680, 4320, 1234, 85, "420", 8)
d = api.Decoder(0)
ts = []
for _ in range(12):
    t = time.perf_counter(); d.read(data, entropy="gpu"); ts.append(time.perf_counter() - t)
print("read ms %.3f" % (min(ts) * 1e3))
PY
for l in ${LANES:-default 1 2 4 8 16 32}; do
  rm -rf /tmp/pl
  if [ "$l" = default ]; then unset MIJPEG_HUFF_LANES; else export MIJPEG_HUFF_LANES=$l; fi
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -o t -- python /tmp/one8k.py > /tmp/pl.log 2>&1
  echo "lanes $l: $(grep 'read ms' /tmp/pl.log)  kernel: $(find /tmp/pl -name '*kernel_stats.csv' -exec grep huffman_scan {} \; | cut -d, -f2-6)"
done
