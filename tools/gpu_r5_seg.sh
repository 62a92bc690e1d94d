#!/bin/bash
# Round 5: the segment size of the self-synchronising first pass on config 5's streams (host only)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5seg
python - > gpurun_out/r5seg/segments.txt 2>&1 <<'PY'
import os, subprocess, sys, tempfile
sys.path.insert(0, os.getcwd())
import bench
from libjpeg_amd import synth
from oracle import oracle as O
W, H = bench.SIZES["4k"]
hdr = synth.synth_hdr(W, H, 99)
d = tempfile.mkdtemp(dir="/dev/shm")
files = {}
for name, extra in (("r12_rR4", ["-rR", "4"]), ("r12", [])):
    files[name] = os.path.join(d, name + ".jpg")
    open(files[name], "wb").write(O.reference_encode_hdr(hdr, bench.XT_ARGS + extra))
child = r'''
import sys, time
sys.path.insert(0, %r)
from libjpeg_amd import api
data = open(sys.argv[1], "rb").read()
d = api.Decoder(None)
ts = []
for i in range(14):
    t = time.perf_counter(); d.read(data, entropy="host"); ts.append((time.perf_counter() - t) * 1e3)
ts = sorted(ts[2:])
print("min %%.2f  median %%.2f ms" %% (ts[0], ts[len(ts) // 2]))
''' % os.getcwd()
open(os.path.join(d, "c.py"), "w").write(child)
for rnd in range(2):
    for seg in (16384, 32768, 49152, 57344, 65536, 114688):
        for name in ("r12", "r12_rR4"):
            r = subprocess.run([sys.executable, os.path.join(d, "c.py"), files[name]], env=dict(os.environ, MIJPEG_SPEC_SEGMENT_BYTES=str(seg)), capture_output=True, text=True)
            print("segment", seg, name, r.stdout.strip(), r.stderr.strip()[-200:], flush=True)
PY
cat gpurun_out/r5seg/segments.txt
