#!/bin/bash
# device walk: parity tests for streams without restart markers + timing + kernel durations per setting
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd $ROOT
[ -n "$SKIP_TESTS" ] || timeout 900 python -m pytest tests -q -x -m gpu -k "without_restart or damaged or batch" 2>&1 | tail -5
cd /tmp; export TMPDIR=/tmp
cat > /tmp/walk_only.py <<PY
import sys, time, os
sys.path.insert(0, "$ROOT")
from libjpeg_amd import api, synth
img = synth.synth_image(7680, 4320, 7)
data = synth.encode_jpeg(img, 85, "420", restart_mcus=0)
d = api.Decoder(0)
ts = []
for it in range(8):
    t0 = time.perf_counter(); d.read(data, entropy="gpu"); ts.append(time.perf_counter() - t0)
print("read ms %.3f rounds %d" % (min(ts) * 1e3, d.device_walk_rounds()), flush=True)
PY
for cfg in ${CFGS:-128:16 128:32 256:16 256:32 512:8 512:16}; do
  set -- ${cfg/:/ }
  export MIJPEG_WALK_SUB=$1 MIJPEG_WALK_LANES=$2
  rm -rf /tmp/pw
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pw -o t -- python /tmp/walk_only.py 2>&1 | grep "read ms" | tr '\n' ' '
  python - <<PY
import csv
r=[x for x in csv.DictReader(open("/tmp/pw/t_kernel_trace.csv"))]
# the last decode: from its first walk round to the scan kernel
idx=[i for i,x in enumerate(r) if "huffman_scan_kernel" in x["Kernel_Name"]][-1]
j=idx
while j>0 and "huffman_scan_kernel" not in r[j-1]["Kernel_Name"]: j-=1
seg=[x for x in r[j:idx+1] if "huffman" in x["Kernel_Name"]]
d=[(int(x["End_Timestamp"])-int(x["Start_Timestamp"]))/1e3 for x in seg]
span=(int(seg[-1]["End_Timestamp"])-int(seg[0]["Start_Timestamp"]))/1e3
print("sub=$1 lanes=$2 kernels us:", [round(v) for v in d], "span us", round(span))
PY
done
unset MIJPEG_WALK_SUB MIJPEG_WALK_LANES
RI=8 timeout 600 python $ROOT/tools/entropy_bench.py 2>&1 | grep "no-DRI 8K to"
