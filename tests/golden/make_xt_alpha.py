#!/usr/bin/env python3
"""Regenerate tests/golden/xt_alpha/: JPEG XT files with an ALPHA CHANNEL, written by the reference's encoder (`jpeg -al alpha.pgm
...`), each with the REAL reference decoder's picture and alpha plane (`jpeg -al alpha_out.pgm in.jpg out.ppm`,
cmd/reconstruct.cpp:154-217, 319-342).  The alpha channel is an image of its own in the ALFA box (Image::ParseAlphaChannel,
codestream/image.cpp:1337-1404) under the alpha merging specification ASPC, with its own residual codestream (ARES) and
refinement boxes (AFIN / ARRF) where the encoder was asked for them; the decoder does not composite.

Cases: 8-bit alpha beside an 8-bit picture (4:4:4 and 4:2:0, progressive), the three compositing methods (-am 1/2/3, matte colour),
-am 0 (opaque: the command line ignores the channel), residual alpha (-ar, -ar12, hidden bits -aR / -arR), 16-bit integer and float
alpha, alpha beside a profile C HDR picture and beside an integer JPEG XT picture.

Run in the build container (needs oracle/_ref/jpeg):   python tests/golden/make_xt_alpha.py
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

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "xt_alpha")
TMP = "/dev/shm" if os.path.isdir("/dev/shm") else None
W, H = 59, 37


def run(args):
    r = subprocess.run([O.REF_BIN, *args], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    assert r.returncode == 0 and b"failed" not in r.stderr, (args, r.stderr)


def read_any(path):
    """-> (samples as '<u2' codes or uint8, is_float): PFM samples go back to the half-float codes they are expansions of"""
    with open(path, "rb") as f:
        magic = f.read(2)
    if magic in (b"PF", b"Pf"):
        f32 = O.read_pfm_reference(path)
        f16 = f32.astype("<f2")
        assert np.array_equal(f16.astype(np.float32).view(np.uint32), f32.view(np.uint32))
        return f16.view("<u2"), True
    return O.read_pnm_any(path), False


def main():
    if not O.have_reference():
        sys.exit("oracle/_ref/jpeg is missing: run `make -C oracle ref` first")
    os.makedirs(OUT, exist_ok=True)
    img = synth.synth_image(W, H, 3)
    a8 = synth.synth_image(W, H, 9, channels=1).reshape(H, W)
    a16 = a8.astype(np.uint16) * 257 + 3
    hdr = synth.synth_hdr(W, H, 5).astype("<f4")
    af = (a8.astype(np.float32) / 255.0).astype("<f4")
    manifest = {}
    with tempfile.TemporaryDirectory(dir=TMP) as d:
        p = lambda n: os.path.join(d, n)  # noqa: E731
        O.write_ppm(p("in.ppm"), img)
        O.write_pfm(p("hdr.pfm"), hdr)
        O.write_ppm(p("a8.pgm"), a8)
        with open(p("a16.pgm"), "wb") as f:
            f.write(b"P5\n%d %d\n65535\n" % (W, H) + a16.astype(">u2").tobytes())
        with open(p("af.pfm"), "wb") as f:
            f.write(b"Pf\n%d %d\n-1.0\n" % (W, H) + af[::-1].tobytes())
        ldr, res = ["-q", "85"], ["-aq", "60", "-aQ", "80", "-ar"]
        cases = {
            "a8": (ldr + ["-al", p("a8.pgm")], "in.ppm"),
            "a8_420": (ldr + ["-al", p("a8.pgm"), "-s", "1x1,2x2,2x2"], "in.ppm"),
            "a8_prog": (ldr + ["-al", p("a8.pgm"), "-v"], "in.ppm"),
            "a8_opaque": (ldr + ["-al", p("a8.pgm"), "-am", "0"], "in.ppm"),
            "a8_premultiplied": (ldr + ["-al", p("a8.pgm"), "-am", "2"], "in.ppm"),
            "a8_matte": (ldr + ["-al", p("a8.pgm"), "-am", "3", "-ab", "10,200,30"], "in.ppm"),
            "a8_residual": (ldr + res + ["-al", p("a8.pgm")], "in.ppm"),
            "a8_residual12": (ldr + ["-aq", "60", "-aQ", "80", "-ar12", "-al", p("a8.pgm")], "in.ppm"),
            "a8_residual_hidden": (ldr + res + ["-aR", "2", "-arR", "1", "-h", "-al", p("a8.pgm")], "in.ppm"),
            "a8_openloop": (ldr + res + ["-aol", "-al", p("a8.pgm")], "in.ppm"),
            "a16_residual": (ldr + ["-aq", "70", "-aQ", "90", "-ar", "-h", "-al", p("a16.pgm")], "in.ppm"),
            "a8_beside_int_xt": (["-r", "-q", "85", "-Q", "90"] + res + ["-al", p("a8.pgm")], "in.ppm"),
            "a8_beside_hdr": (["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c", "-r12", "-al", p("a8.pgm")], "hdr.pfm"),
            "af_beside_hdr": (["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c", "-r12", "-aq", "70", "-aQ", "90", "-ar", "-al", p("af.pfm")], "hdr.pfm"),
        }
        for name, (args, src) in cases.items():
            run(args + [p(src), p("o.jpg")])
            blob = open(p("o.jpg"), "rb").read()
            assert b"ALFA" in blob, name
            for f in ("alpha.out", "pic.out"):
                if os.path.exists(p(f)):
                    os.remove(p(f))
            run(["-al", p("alpha.out"), p("o.jpg"), p("pic.out")])
            pic, pic_float = read_any(p("pic.out"))
            ent = dict(jpeg_sha256=hashlib.sha256(blob).hexdigest(), width=W, height=H, picture_float=pic_float, picture_dtype=pic.dtype.str)
            pic.reshape(H, W, 3).tofile(os.path.join(OUT, name + ".pic.bin"))
            if os.path.exists(p("alpha.out")):
                alpha, alpha_float = read_any(p("alpha.out"))
                alpha.reshape(H, W).tofile(os.path.join(OUT, name + ".alpha.bin"))
                ent.update(alpha_float=alpha_float, alpha_dtype=alpha.dtype.str)
            else:  # compositing method 0: the reference's command line writes no alpha file (cmd/reconstruct.cpp:154-166)
                assert name == "a8_opaque"
                ent.update(alpha_float=False, alpha_dtype=None)
            with open(os.path.join(OUT, name + ".jpg"), "wb") as f:
                f.write(blob)
            manifest[name] = ent
            print(f"{name:22s} {len(blob):6d} bytes, alpha {ent['alpha_dtype']} float={ent['alpha_float']}")
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
