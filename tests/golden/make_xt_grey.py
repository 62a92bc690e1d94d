#!/usr/bin/env python3
"""Regenerate tests/golden/xt_grey/: JPEG XT (profile C) files with ONE component -- a grey scale picture with a residual
codestream, what the reference encoder writes for `jpeg -r -q .. -Q .. in.pgm` (8- and 16-bit samples) and
`jpeg -r .. -h -profile c in.pfm` with a one-channel PFM -- and the REAL reference decoder's output for each (PGM samples or the
float32 of its PFM).  Every transformation is the identity there (codestream/tables.cpp:2003-2005, 2055-2060, 2079-2081;
YCbCrTrafo<.., 1, .., Identity, Identity>, colortrafo/colortransformerfactory.cpp:681-757).

Run in the build container (needs oracle/_ref/jpeg):   python tests/golden/make_xt_grey.py
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

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "xt_grey")
TMP = "/dev/shm" if os.path.isdir("/dev/shm") else None


def encode(header, raw, ext, args):
    with tempfile.TemporaryDirectory(dir=TMP) as d:
        src, dst = os.path.join(d, "in." + ext), os.path.join(d, "out.jpg")
        with open(src, "wb") as f:
            f.write(header + raw)
        subprocess.run([O.REF_BIN, *args, src, dst], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        with open(dst, "rb") as f:
            return f.read()


def decode(data, ext):
    with tempfile.TemporaryDirectory(dir=TMP) as d:
        src, dst = os.path.join(d, "in.jpg"), os.path.join(d, "out." + ext)
        with open(src, "wb") as f:
            f.write(data)
        subprocess.run([O.REF_BIN, src, dst], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        return O.read_pfm_reference(dst).astype("<f4") if ext == "pfm" else O.read_pnm_any(dst)


def main():
    if not O.have_reference():
        sys.exit("oracle/_ref/jpeg is missing: run `make -C oracle ref` first")
    os.makedirs(OUT, exist_ok=True)
    w, h = 83, 45
    g = synth.synth_image(w, h, 7, channels=1).reshape(h, w)
    g16 = g.astype(np.uint16) * 257 + 13
    ghdr = (synth.synth_hdr(w, h, 5)[:, :, 1] * 4.0).astype("<f4")
    pgm = (b"P5\n%d %d\n255\n" % (w, h), g.tobytes(), "pgm")
    pgm16 = (b"P5\n%d %d\n65535\n" % (w, h), g16.astype(">u2").tobytes(), "pgm")
    pfm = (b"Pf\n%d %d\n-1.0\n" % (w, h), ghdr[::-1].tobytes(), "pfm")
    ldr, hdr = ["-r", "-q", "85", "-Q", "90"], ["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c", "-r12"]
    cases = {
        "g8": encode(*pgm, ldr), "g8_r12": encode(*pgm, ldr + ["-r12"]), "g8_rR2": encode(*pgm, ldr + ["-rR", "2"]),
        "g8_prog_q60": encode(*pgm, ["-r", "-q", "60", "-Q", "70", "-v"]), "g16": encode(*pgm16, ldr),
        "ghdr": encode(*pfm, hdr), "ghdr_R1_rR3": encode(*pfm, hdr + ["-rR", "3", "-R", "1"]),
    }
    manifest = {}
    for name, blob in cases.items():
        assert blob, name
        with open(os.path.join(OUT, name + ".jpg"), "wb") as f:
            f.write(blob)
        px = decode(blob, "pfm" if name.startswith("ghdr") else "pgm")
        px = px.reshape(h, w)
        px.tofile(os.path.join(OUT, name + ".bin"))
        manifest[name] = dict(jpeg_sha256=hashlib.sha256(blob).hexdigest(), width=w, height=h, dtype=px.dtype.str,
                              pixels_sha256=hashlib.sha256(px.tobytes()).hexdigest())
        print(f"{name:16s} {px.dtype}")
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
