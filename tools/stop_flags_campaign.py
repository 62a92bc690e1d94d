#!/usr/bin/env python3
"""CPU only, build container (needs the reference's objects under oracle/_ref/obj): random VALID streams -- every sampling layout,
grey, progressive, restart intervals, 12-bit, JPEG XT goldens -- through tests/cxx/marker_calls.cpp built against this library and
against the reference library: the traces of JPGFLAG_DECODER_STOP_IMAGE / _FRAME / _SCAN loops line by line, the number of returns of
the _ROW / _MCU loops (what tests/test_marker_calls.py checks on the goldens).    N=300 SEED=1 [FILES=every] python tools/stop_flags_campaign.py"""
import collections
import glob
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libjpeg_amd import synth  # noqa: E402

SRC = os.path.join(ROOT, "tests", "cxx", "marker_calls.cpp")
N, SEED = int(os.environ.get("N", "300")), int(os.environ.get("SEED", "1"))


def build(d):
    ours, ref = os.path.join(d, "ours"), os.path.join(d, "ref")
    subprocess.run(["g++", "-O1", "-w", "-I", os.path.join(ROOT, "libjpeg_amd", "csrc"), SRC, "-o", ours, "-L", os.path.join(ROOT, "libjpeg_amd"),
                    "-lmijpeg", "-Wl,-rpath," + os.path.join(ROOT, "libjpeg_amd")], check=True)
    objs = [o for o in glob.glob(os.path.join(ROOT, "oracle", "_ref", "obj", "**", "*.o"), recursive=True) if os.sep + "cmd" + os.sep not in o]
    refsrc = os.environ.get("LIBJPEG_REFERENCE", "/root/reference")
    subprocess.run(["g++", "-O1", "-w", "-DUSE_AUTOCONF", "-fno-exceptions", "-I", refsrc, "-I", os.path.join(ROOT, "oracle", "_ref", "gen"), SRC, *objs, "-o", ref, "-lm"], check=True)
    return ours, ref


def run(exe, path, mode):
    r = subprocess.run([exe, path, mode, "0"], capture_output=True, text=True, timeout=120, env=dict(os.environ, MIJPEG_DEVICE="-1"))
    return r.returncode, r.stdout.strip().splitlines()


def main():
    rng = np.random.default_rng(SEED)
    count = collections.Counter()
    bad = []
    # (JPEG XT: the profile C goldens, alpha channels included; FILES=every: all classes and the DNL frames -- what is left to differ
    # there are files whose colour transformer refuses: the reference says so at the first request for pixels, this library at the
    # read)
    xt = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "xt_*.jpg")) + glob.glob(os.path.join(ROOT, "tests", "golden", "xt_grey", "g*_r12.jpg")) +
                glob.glob(os.path.join(ROOT, "tests", "golden", "xt_alpha", "*.jpg")))
    if os.environ.get("FILES") == "every":  # every JPEG XT golden class, and DNL frames
        xt = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "xt_*.jpg")) + glob.glob(os.path.join(ROOT, "tests", "golden", "xt_*", "*.jpg")) +
                    glob.glob(os.path.join(ROOT, "tests", "golden", "dnl", "*.jpg")))
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
        ours, ref = build(d)
        for i in range(N):
            path = os.path.join(d, "in.jpg")
            if i % 5 == 4 or os.environ.get("FILES") == "every":
                src = xt[int(rng.integers(0, len(xt)))]
                data = open(src, "rb").read()
                what = os.path.basename(src)
            else:
                w, h = int(rng.integers(1, 200)), int(rng.integers(1, 150))
                sub = ["444", "422", "420", "gray", "440", "411"][int(rng.integers(0, 6))]
                dri = int(rng.choice([0, 0, 1, 3, 8]))
                prog = bool(rng.integers(0, 3) == 0)
                img = synth.synth_image(w, h, SEED * 100000 + i, channels=1 if sub == "gray" else 3)
                try:
                    data = synth.encode_jpeg(img, int(rng.choice([30, 75, 95])), sub if sub != "gray" else "444", restart_mcus=dri, progressive=prog)
                    if rng.integers(0, 5) == 0:
                        data = synth.to_12bit(data)
                except Exception:  # noqa: BLE001
                    count["refused by the test encoder"] += 1
                    continue
                what = f"{w}x{h} {sub} dri{dri} {'prog' if prog else 'seq'}"
            open(path, "wb").write(data)
            for mode in ("image", "frame", "scan", "row", "mcu", "scanrow"):
                a, b = run(ours, path, mode), run(ref, path, mode)
                if a == b:
                    count["same " + mode] += 1
                else:
                    count["MISMATCH " + mode] += 1
                    bad.append((i, what, mode, a[0], b[0], a[1][:3], b[1][:3]))
    print("seed", SEED, dict(count))
    for x in bad[:30]:
        print("MISMATCH", x)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
