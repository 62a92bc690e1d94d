#!/bin/bash
# kernel durations of the device walk over a batch of no-DRI 8K streams
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp; export TMPDIR=/tmp
cat > /tmp/walk_batch.py <<PY
import sys, time, os
sys.path.insert(0, "$ROOT")
from libjpeg_amd import api, synth
N = int(os.environ.get("NFRAMES", "16"))
frames = [synth.encode_jpeg(synth.synth_image(7680, 4320, 2000 + i), 85, "420", restart_mcus=0) for i in range(4)]
batch = [frames[i % 4] for i in range(N)]
d = api.Decoder(0)
ts = []
for it in range(4):
    t0 = time.perf_counter(); d.decode_batch_device(batch); ts.append(time.perf_counter() - t0)
print("decode_batch ms %.3f rounds %d" % (min(ts) * 1e3, d.device_walk_rounds()), flush=True)
PY
rm -rf /tmp/pw
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pw -o t -- python /tmp/walk_batch.py 2>&1 | grep "decode_batch"
python - <<PY
import csv
r=[x for x in csv.DictReader(open("/tmp/pw/t_kernel_trace.csv"))]
last=lambda x: "huffman_scan_kernel" in x["Kernel_Name"] or "huffman_subscan_kernel" in x["Kernel_Name"]
idx=[i for i,x in enumerate(r) if last(x)][-1]
j=idx
while j>0 and not last(r[j-1]): j-=1
seg=[x for x in r[j:idx+1] if "huffman" in x["Kernel_Name"]]
t0=int(seg[0]["Start_Timestamp"])
for x in seg:
    s,e=int(x["Start_Timestamp"]),int(x["End_Timestamp"])
    print("%-44s start %8.1f dur %8.1f us grid %s wg %s" % (x["Kernel_Name"][5:49], (s-t0)/1e3, (e-s)/1e3, x["Grid_Size_X"], x["Workgroup_Size_X"]))
PY
