#!/usr/bin/env python3
"""Regenerate tests/golden/xt_int16/: JPEG XT (profile C) files with a 16-bit INTEGER output -- what the reference encoder writes
for `jpeg -r -q .. -Q .. in.ppm` when in.ppm has 16-bit samples (output conversion with 8 extra range bits, clamping, no cast
to float) -- with the REAL reference decoder's PPM samples, plain and with `-c`.

Run in the build container (needs oracle/_ref/jpeg):   python tests/golden/make_xt_int16.py
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from libjpeg_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "xt_int16")
TMP = "/dev/shm" if os.path.isdir("/dev/shm") else None


def encode16(img, args):
    with tempfile.TemporaryDirectory(dir=TMP) as d:
        src, dst = os.path.join(d, "in.ppm"), os.path.join(d, "out.jpg")
        with open(src, "wb") as f:
            f.write(b"P6\n%d %d\n65535\n" % (img.shape[1], img.shape[0]))
            f.write(img.astype(">u2").tobytes())
        subprocess.run([O.REF_BIN, *args, src, dst], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        with open(dst, "rb") as f:
            return f.read()


def decode(data, extra):
    with tempfile.TemporaryDirectory(dir=TMP) as d:
        src, dst = os.path.join(d, "in.jpg"), os.path.join(d, "out.ppm")
        with open(src, "wb") as f:
            f.write(data)
        subprocess.run([O.REF_BIN, *extra, src, dst], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return O.read_pnm_any(dst)


def main():
    if not O.have_reference():
        sys.exit("oracle/_ref/jpeg is missing: run `make -C oracle ref` first")
    os.makedirs(OUT, exist_ok=True)
    img = synth.synth_image(83, 45, 7).astype(np.uint16) * 257 + 13
    base = ["-r", "-q", "85", "-Q", "90"]
    cases = {"w444": encode16(img, base), "w420": encode16(img, base + ["-s", "1x1,2x2,2x2"]), "w420_r12": encode16(img, base + ["-r12", "-s", "1x1,2x2,2x2"]),
             "wxyz": encode16(img, base + ["-xyz"])}
    manifest = {}
    for name, blob in cases.items():
        assert blob, name
        with open(os.path.join(OUT, name + ".jpg"), "wb") as f:
            f.write(blob)
        ent = dict(jpeg_sha256=hashlib.sha256(blob).hexdigest())
        for tag, extra in (("plain", []), ("noct", ["-c"])):
            px = decode(blob, extra)
            assert px.dtype == np.uint16, (name, px.dtype)
            px.astype("<u2").tofile(os.path.join(OUT, f"{name}.{tag}.bin"))
            ent.update(height=int(px.shape[0]), width=int(px.shape[1]))
            ent[tag + "_sha256"] = hashlib.sha256(px.astype("<u2").tobytes()).hexdigest()
        manifest[name] = ent
        print(name, "written")
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
