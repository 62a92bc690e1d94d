#!/usr/bin/env python3
"""GPU box: the device decoder of progressive frames / hidden refinement scans (huffman_prog_kernel) on config 5's `-rR 4 -z 8`
stream and on progressive pictures with restart markers -- read times host against device and the host-side steps of the device
path (MIJPEG_TRACE_SUBMIT=1, MIJPEG_READ_TIMES=1); under rocprofv3 --kernel-trace --stats the kernels themselves.
  python tools/multiscan_probe.py make DIR            the streams, written by the reference encoder (not under a profiler)
  python tools/multiscan_probe.py run DIR [reads] [names..]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libjpeg_amd import api, synth  # noqa: E402

mode, where = sys.argv[1], sys.argv[2]
if mode == "make":
    from oracle import oracle as O

    os.makedirs(where, exist_ok=True)
    base = ["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c", "-r12", "-s", "1x1,2x2,2x2"]
    hdr = synth.synth_hdr(3840, 2160, 99)
    out = {"xt4k_rR4_z8": O.reference_encode_hdr(hdr, base + ["-rR", "4", "-z", "8"]),
           "prog4k_z8": O.reference_encode(synth.synth_image(3840, 2160, 77), ["-v", "-q", "85", "-s", "1x1,2x2,2x2", "-z", "8"]),
           "prog8k_z8": O.reference_encode(synth.synth_image(7680, 4320, 1234), ["-v", "-q", "85", "-s", "1x1,2x2,2x2", "-z", "8"])}
    for k, v in out.items():
        with open(os.path.join(where, k + ".jpg"), "wb") as f:
            f.write(v)
        print(k, len(v))
    sys.exit(0)
reads = int(sys.argv[3]) if len(sys.argv) > 3 else 5
names = sys.argv[4:] or sorted(n[:-4] for n in os.listdir(where) if n.endswith(".jpg"))
dec = api.Decoder(0)
for name in names:
    with open(os.path.join(where, name + ".jpg"), "rb") as f:
        data = f.read()
    for m in ("host", "prefer-gpu"):
        ts = []
        for _ in range(reads):
            t = time.perf_counter()
            dec.read(data, entropy=m)
            ts.append((time.perf_counter() - t) * 1e3)
        print(f"{name:16s} {len(data):9d} bytes  {m:10s} ran on {dec.entropy_used:4s}  min {min(ts):7.2f} ms  all {[round(x, 2) for x in ts]}", flush=True)
dec.close()
