#!/bin/bash
# kernel durations of the device walk (no-DRI 8K stream)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp; export TMPDIR=/tmp
cat > /tmp/walk_only.py <<PY
import sys, time, os
sys.path.insert(0, "$ROOT")
from libjpeg_amd import api, synth
img = synth.synth_image(7680, 4320, 7)
data = synth.encode_jpeg(img, 85, "420", restart_mcus=0)
d = api.Decoder(0)
for it in range(8):
    t0 = time.perf_counter(); d.read(data, entropy="gpu"); t1 = time.perf_counter()
    print("read ms", (t1 - t0) * 1e3, d.device_walk_rounds(), flush=True)
PY
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/prof_walk -o t -- python /tmp/walk_only.py 2>&1 | grep "read ms"
python - <<PY
import csv
r=[x for x in csv.DictReader(open("$ROOT/gpurun_out/prof_walk/t_kernel_trace.csv"))]
t0=None
for x in r[-8:]:
    s,e=int(x["Start_Timestamp"]),int(x["End_Timestamp"])
    if t0 is None: t0=s
    print(x["Kernel_Name"][:40], "start", (s-t0)/1e3, "dur us", (e-s)/1e3, "grid", x["Grid_Size"], "wg", x["Workgroup_Size"], "lds", x["LDS_Block_Size"], "vgpr", x["VGPR_Count"])
PY
