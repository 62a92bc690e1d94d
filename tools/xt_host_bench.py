#!/usr/bin/env python3
"""Host entropy decode of BASELINE config 5's `-rR 4` variant (4K HDR, 12-bit residual with four hidden bits: one base scan
and 16 hidden refinement scans in RFIN boxes) with the switches of libjpeg_amd/csrc/host_decoder.cpp, each variant in a
process of its own: the mask-based AC refinement pass (MIJPEG_NO_REFINE_MASKS), the self-synchronising parallel decode of the
first full-band pass (MIJPEG_NO_SPEC_FIRST_PASS), the scan pipeline (MIJPEG_NO_SCAN_PIPELINE).  Prints best-of-N milliseconds.

    python tools/xt_host_bench.py [threads ...]          (needs oracle/_ref/jpeg for the encoder; caches the stream in /dev/shm)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STREAM = "/dev/shm/mijpeg_rr4_4k.jpg"

CHILD = r'''
import sys, time
sys.path.insert(0, %r)
from libjpeg_amd import api
data = open(%r, "rb").read()
d = api.Decoder(None)
for th in %r:
    ts = []
    for _ in range(7):
        t = time.perf_counter(); d.read(data, threads=th); ts.append((time.perf_counter() - t) * 1e3)
    print("    threads %%3d: best %%7.1f ms   median %%7.1f ms" %% (th, min(ts), sorted(ts)[len(ts) // 2]))
'''


def main():
    threads = [int(x) for x in sys.argv[1:]] or [1, 8, 64]
    if not os.path.exists(STREAM):
        from libjpeg_amd import synth
        from oracle import oracle as O
        data = O.reference_encode_hdr(synth.synth_hdr(3840, 2160, 99), ["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c", "-r12", "-s", "1x1,2x2,2x2", "-rR", "4"])
        with open(STREAM, "wb") as f:
            f.write(data)
    print("stream:", os.path.getsize(STREAM), "bytes;", os.cpu_count(), "cores")
    variants = [("as built", {}), ("no mask refinement", {"MIJPEG_NO_REFINE_MASKS": "1"}), ("no parallel first pass", {"MIJPEG_NO_SPEC_FIRST_PASS": "1"}),
                ("neither (round 3)", {"MIJPEG_NO_REFINE_MASKS": "1", "MIJPEG_NO_SPEC_FIRST_PASS": "1"}),
                ("no scan pipeline", {"MIJPEG_NO_SCAN_PIPELINE": "1"})]
    for name, env in variants:
        print(name)
        r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, STREAM, threads)], env=dict(os.environ, **env), capture_output=True, text=True)
        print(r.stdout.rstrip() or r.stderr[-400:])


if __name__ == "__main__":
    main()
