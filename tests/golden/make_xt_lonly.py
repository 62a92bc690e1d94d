#!/usr/bin/env python3
"""Regenerate tests/golden/xt_lonly/: files with a merging specification and NO residual codestream whose specification asks
for more than the plain picture -- what the reference's encoder writes for `jpeg -q .. -R n -h in` (no `-r`) from a picture of
more than eight bits: an 8-bit legacy codestream, n hidden refinement bits in FINE boxes, a TONE box with the L table from
8 + n bits to 16, an output conversion (cast to half float for PFM input, 16-bit integers for 16-bit PNM input).  The reference
decodes them through the L chain alone (colortrafo/colortransformerfactory.cpp:262-283, 300-594 with `residual` false;
colortrafo/ycbcrtrafo.cpp:744-746, 861-878).  Each stream comes with the REAL reference decoder's output: the half-float codes
behind its PFM (its float32 samples are exact expansions of them, cmd/iohelpers.hpp:60-77; checked here) or its 16-bit PNM samples.

{PFM, one-channel PFM, 16-bit PPM, 16-bit PGM} x -R 1..4 x {sequential, progressive -v, restart intervals -z 2,
4:2:0 -s 1x1,2x2,2x2 (three components only)}.

Run in the build container (needs oracle/_ref/jpeg):   python tests/golden/make_xt_lonly.py
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

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "xt_lonly")
TMP = "/dev/shm" if os.path.isdir("/dev/shm") else None
W, H = 43, 29


def encode(header, raw, ext, args):
    with tempfile.TemporaryDirectory(dir=TMP) as d:
        src, dst = os.path.join(d, "in." + ext), os.path.join(d, "out.jpg")
        with open(src, "wb") as f:
            f.write(header + raw)
        r = subprocess.run([O.REF_BIN, *args, src, dst], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        assert r.returncode == 0 and b"failed" not in r.stderr, (args, r.stderr)
        with open(dst, "rb") as f:
            return f.read()


def decode(data, ext):
    with tempfile.TemporaryDirectory(dir=TMP) as d:
        src, dst = os.path.join(d, "in.jpg"), os.path.join(d, "out." + ext)
        with open(src, "wb") as f:
            f.write(data)
        r = subprocess.run([O.REF_BIN, src, dst], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        assert r.returncode == 0 and b"failed" not in r.stderr, r.stderr
        if ext == "pfm":
            f32 = O.read_pfm_reference(dst)
            f16 = f32.astype("<f2")
            assert np.array_equal(f16.astype(np.float32).view(np.uint32), f32.view(np.uint32))  # exact expansions of half codes
            return f16.view("<u2")
        return O.read_pnm_any(dst).astype("<u2")


def sources(W=W, H=H):
    hdr = synth.synth_hdr(W, H, 5).astype("<f4")
    i16 = (np.clip(hdr / hdr.max(), 0, 1) ** 0.45 * 65535).astype(np.uint16)
    return {
        "hdr": (b"PF\n%d %d\n-1.0\n" % (W, H), hdr[::-1].tobytes(), "pfm", 3),
        "ghdr": (b"Pf\n%d %d\n-1.0\n" % (W, H), (hdr[:, :, 1] * 4.0).astype("<f4")[::-1].tobytes(), "pfm", 1),
        "i16": (b"P6\n%d %d\n65535\n" % (W, H), i16.astype(">u2").tobytes(), "ppm", 3),
        "g16": (b"P5\n%d %d\n65535\n" % (W, H), i16[:, :, 1].astype(">u2").tobytes(), "pgm", 1),
    }


MODES = {"seq": [], "prog": ["-v"], "z2": ["-z", "2"], "420": ["-s", "1x1,2x2,2x2"]}


def main():
    if not O.have_reference():
        sys.exit("oracle/_ref/jpeg is missing: run `make -C oracle ref` first")
    os.makedirs(OUT, exist_ok=True)
    manifest = {}
    for sname, (header, raw, ext, ch) in sources().items():
        for hidden in (1, 2, 3, 4):
            for mname, margs in MODES.items():
                if mname == "420" and ch == 1:
                    continue
                name = f"{sname}_R{hidden}_{mname}"
                blob = encode(header, raw, ext, ["-q", "85", "-R", str(hidden), "-h", *margs])
                assert b"RESI" not in blob and b"SPEC" in blob, name
                px = decode(blob, "pfm" if ext == "pfm" else ext).reshape(H, W, ch)
                with open(os.path.join(OUT, name + ".jpg"), "wb") as f:
                    f.write(blob)
                px.tofile(os.path.join(OUT, name + ".bin"))
                manifest[name] = dict(jpeg_sha256=hashlib.sha256(blob).hexdigest(), width=W, height=H, channels=ch,
                                      is_float=ext == "pfm", hidden=hidden, pixels_sha256=hashlib.sha256(px.tobytes()).hexdigest())
                print(f"{name:18s} {len(blob):6d} bytes")
    # Frames whose height arrives in a DNL marker (`-n`) -- legal here, where no residual codestream's frame header would disagree
    # with the legacy one's: the reference's rows behind the picture and upsamplers without a bottom edge under the L
    # chain.  Heights on and off the block grid.
    for h in (29, 32):
        for sname, (header, raw, ext, ch) in sources(W, h).items():
            for mname, margs in (("dnl", ["-n"]), ("dnl420", ["-n", "-s", "1x1,2x2,2x2"]), ("dnlprog", ["-n", "-v"])):
                if mname == "dnl420" and ch == 1:
                    continue
                name = f"{sname}_R2_{mname}_h{h}"
                blob = encode(header, raw, ext, ["-q", "85", "-R", "2", "-h", *margs])
                assert b"RESI" not in blob and b"SPEC" in blob and b"\xff\xdc\x00\x04" in blob, name
                px = decode(blob, "pfm" if ext == "pfm" else ext).reshape(h, W, ch)
                with open(os.path.join(OUT, name + ".jpg"), "wb") as f:
                    f.write(blob)
                px.tofile(os.path.join(OUT, name + ".bin"))
                manifest[name] = dict(jpeg_sha256=hashlib.sha256(blob).hexdigest(), width=W, height=h, channels=ch,
                                      is_float=ext == "pfm", hidden=2, pixels_sha256=hashlib.sha256(px.tobytes()).hexdigest())
                print(f"{name:24s} {len(blob):6d} bytes")
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
