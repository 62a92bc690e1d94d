#!/usr/bin/env python3
"""Regenerate tests/golden/xt_lossless/: JPEG XT files of the lossless / near-lossless kind (part 8), written by the reference's encoder
with `-ro` (no DCT in the residual domain: spatial quantisation) or `-Q 100`, each with the REAL reference decoder's output.  Their RESI
box holds a codestream of the RESIDUAL scan type (SOF 0xffb1: no DC coding, the value -0x8000 as symbol 0x10; marker/scan.cpp:483-489,
codestream/sequentialscan.cpp:678-773) of 8 + 1 or 16 + 1 bits; the merging specification names the RCT as R transformation (three
components) or none (one component: the identity), the DCT bypass, and an output conversion with the lossless flag and WITHOUT clamping:
the transformers YCbCrTrafo<.., Residual | Extended [| Float], .., RCT / Identity> (colortrafo/colortransformerfactory.cpp:726-757,
826-849, 963-990; colortrafo/ycbcrtrafo.cpp:752-766, 797-801, 820-822, 940-972).

Run in the build container (needs oracle/_ref/jpeg):   python tests/golden/make_xt_lossless.py
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

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "xt_lossless")
TMP = "/dev/shm" if os.path.isdir("/dev/shm") else None
W, H = 53, 37


def main():
    if not O.have_reference():
        sys.exit("oracle/_ref/jpeg is missing: run `make -C oracle ref` first")
    os.makedirs(OUT, exist_ok=True)
    img = synth.synth_image(W, H, 3)
    hdr = synth.synth_hdr(W, H, 5).astype("<f4")
    i16 = (np.clip(hdr / hdr.max(), 0, 1) ** 0.45 * 65535).astype(np.uint16)
    manifest = {}
    with tempfile.TemporaryDirectory(dir=TMP) as d:
        p = lambda n: os.path.join(d, n)  # noqa: E731
        O.write_ppm(p("rgb8.ppm"), img)
        O.write_ppm(p("grey8.pgm"), img[:, :, 1])
        O.write_pfm(p("hdr.pfm"), hdr)
        with open(p("ghdr.pfm"), "wb") as f:
            f.write(b"Pf\n%d %d\n-1.0\n" % (W, H) + hdr[:, :, 1][::-1].tobytes())
        with open(p("rgb16.ppm"), "wb") as f:
            f.write(b"P6\n%d %d\n65535\n" % (W, H) + i16.astype(">u2").tobytes())
        with open(p("grey16.pgm"), "wb") as f:
            f.write(b"P5\n%d %d\n65535\n" % (W, H) + i16[:, :, 2].astype(">u2").tobytes())
        ro = ["-r", "-q", "85", "-Q", "90", "-ro", "-h"]
        cases = {
            "rgb8_ro": ("rgb8.ppm", ro), "rgb8_q100": ("rgb8.ppm", ["-r", "-q", "85", "-Q", "100", "-h"]), "rgb8_ro_q30": ("rgb8.ppm", ["-r", "-q", "30", "-Q", "60", "-ro", "-h"]),
            "rgb8_ro_420": ("rgb8.ppm", ro + ["-s", "1x1,2x2,2x2"]), "rgb8_ro_noise": ("rgb8.ppm", ro + ["-N"]), "rgb8_ro_noct": ("rgb8.ppm", ro + ["-c"]),
            "rgb8_ro_dri4": ("rgb8.ppm", ro + ["-z", "4"]), "rgb8_ro_prog": ("rgb8.ppm", ro + ["-v"]),
            "grey8_ro": ("grey8.pgm", ro), "rgb16_ro": ("rgb16.ppm", ro), "grey16_ro": ("grey16.pgm", ro),
            "hdr_ro": ("hdr.pfm", ro + ["-profile", "c"]), "ghdr_ro": ("ghdr.pfm", ro + ["-profile", "c"]),
            # round 5, second half: the progressive residual type (SOF 0xffb2, -rv) and hidden refinement scans of the residual kind (-rR n)
            "rgb8_ro_rv": ("rgb8.ppm", ro + ["-rv"]), "rgb8_ro_rR2": ("rgb8.ppm", ro + ["-rR", "2"]),
            "rgb8_q100_rv_rR3": ("rgb8.ppm", ["-r", "-q", "85", "-Q", "100", "-h", "-rv", "-rR", "3"]),
            "rgb8_ro_rv_rR2_420_dri4": ("rgb8.ppm", ro + ["-rv", "-rR", "2", "-s", "1x1,2x2,2x2", "-z", "4"]),
            "grey16_ro_rv": ("grey16.pgm", ro + ["-rv"]), "rgb16_ro_rR4": ("rgb16.ppm", ro + ["-rR", "4"]), "hdr_ro_rv_rR1": ("hdr.pfm", ro + ["-profile", "c", "-rv", "-rR", "1"]),
        }
        for name, (src, args) in cases.items():
            r = subprocess.run([O.REF_BIN, *args, p(src), p("o.jpg")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            assert r.returncode == 0 and b"failed" not in r.stderr, (name, r.stderr)
            blob = open(p("o.jpg"), "rb").read()
            assert b"RESI" in blob, name
            r = subprocess.run([O.REF_BIN, p("o.jpg"), p("o.out")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            assert r.returncode == 0 and b"failed" not in r.stderr, (name, r.stderr)
            magic = open(p("o.out"), "rb").read(2)
            if magic in (b"PF", b"Pf"):
                f32 = O.read_pfm_reference(p("o.out"))
                px = f32.astype("<f2")
                assert np.array_equal(px.astype(np.float32).view(np.uint32), f32.view(np.uint32))
                px, is_float = px.view("<u2"), True
            else:
                px, is_float = O.read_pnm_any(p("o.out")), False
            ch = 1 if magic in (b"Pf", b"P5") else 3
            px = px.reshape(H, W, ch)
            with open(os.path.join(OUT, name + ".jpg"), "wb") as f:
                f.write(blob)
            px.tofile(os.path.join(OUT, name + ".bin"))
            manifest[name] = dict(jpeg_sha256=hashlib.sha256(blob).hexdigest(), width=W, height=H, channels=ch, dtype=px.dtype.str, is_float=is_float,
                                  pixels_sha256=hashlib.sha256(px.tobytes()).hexdigest())
            print(f"{name:16s} {len(blob):6d} bytes  {px.dtype} float={is_float}")
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
