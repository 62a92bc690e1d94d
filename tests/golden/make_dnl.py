#!/usr/bin/env python3
"""Regenerate tests/golden/dnl/: frames whose height arrives in a DNL marker, and what the REAL reference decodes them to.

Run in the build container (needs oracle/_ref/jpeg):   python tests/golden/make_dnl.py

The reference sets up its block rows and its upsamplers while the height is still unknown, and it looks for the marker in
the byte stream while its bit reader has read ahead (DESIGN.md "DNL frames"): the last line(s) of vertically subsampled
components and the last MCUs of the first scan come out differently from the same frame with the height in its header.
The cases: the reference encoder's -n over samplings x heights (block-row aligned, one off, ragged) x {baseline,
progressive, restart intervals}; flat pictures, whose MCUs are so short that several of them sit in the bit reader's
window when the marker is seen; hand-written frames of two and four components (tests/craft.py), which the command line
reconstructs component by component.  Stored: <name>.jpg + sha256 of the pixel bytes the reference wrote.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import craft  # noqa: E402
from libjpeg_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dnl")

SAMPLINGS = {"420": "1x1,2x2,2x2", "1x2": "1x1,1x2,1x2", "lumasub": "2x2,1x1,1x1", "1x3": "1x1,1x3,1x3", "1x4": "1x1,1x4,1x4",
             "mixed": "1x1,2x2,1x1", "422": "1x1,2x1,2x1", "444": "1x1,1x1,1x1"}
HEIGHTS = [16, 32, 33, 46, 48, 70]
MODES = {"bl": ["-bl"], "prog": ["-v"], "dri2": ["-z", "2"]}


def sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def main():
    O.build()
    assert O.have_reference(), "oracle/_ref/jpeg missing: run `make -C oracle ref`"
    os.makedirs(OUT, exist_ok=True)
    manifest = {}

    def add(name, data, **meta):
        px, err = O.reference_decode_status(data)
        assert err == 0 and px is not None, (name, err)
        with open(os.path.join(OUT, name + ".jpg"), "wb") as f:
            f.write(data)
        manifest[name] = dict(meta, height=int(px.shape[0]), width=int(px.shape[1]), channels=int(px.shape[2]), dtype=str(px.dtype),
                              jpeg_sha256=sha(data), pixels_sha256=sha(np.ascontiguousarray(px).tobytes()))

    for sn, s in SAMPLINGS.items():
        for h in HEIGHTS:
            for mn, m in MODES.items():
                args = m + ["-q", "85", "-n", "-s", s]
                add(f"dnl_{sn}_{h}_{mn}", O.reference_encode(synth.synth_image(80, h, 1234), args), kind="ref", args=args)
    # flat and two-level pictures at several qualities: MCUs of a few bits
    rng = np.random.default_rng(2024)
    for k in range(16):
        w, h = int(rng.integers(8, 100)), int(rng.choice([16, 24, 32, 40, 48, 64]))
        img = np.full((h, w, 3), int(rng.integers(256)), np.uint8)
        if k & 1:
            img[: h // 2] = 40
        s = list(SAMPLINGS.values())[k % len(SAMPLINGS)]
        args = [["-bl"], ["-v"], ["-z", "3"], ["-h"]][k % 4] + ["-q", str(int(rng.choice([5, 50, 95]))), "-n", "-s", s]
        add(f"dnl_flat_{k}", O.reference_encode(img, args), kind="ref-flat", args=args)
    # grey scale
    for k, h in enumerate([16, 33, 48]):
        args = [["-bl"], ["-v"], ["-z", "2"]][k] + ["-q", "80", "-n"]
        add(f"dnl_gray_{h}", O.reference_encode(synth.synth_image(70, h, 99)[:, :, :1], args), kind="ref", args=args)
    # hand-written frames of 1..4 components with mixed sampling (no encoder writes those)
    n = 0
    while n < 24:
        nc = int(rng.integers(2, 5))
        samp = [(int(rng.integers(1, 5)), int(rng.integers(1, 5))) for _ in range(nc)]
        hm, vm = max(s[0] for s in samp), max(s[1] for s in samp)
        if any(hm % s[0] or vm % s[1] for s in samp) or sum(s[0] * s[1] for s in samp) > 10:
            continue
        w, h = int(rng.integers(1, 100)), int(rng.choice([8, 16, 24, 31, 32, 48, 56, 64, 77]))
        dri = int(rng.choice([0, 0, 1, 3, 7]))
        dens = float(rng.choice([0.0, 0.02, 0.08, 0.3]))
        data = craft.to_dnl(craft.craft_stream(rng, samp, w, h, dri=dri, ac_density=dens))
        add(f"dnl_craft_{n}", data, kind="craft", samp=samp, dri=dri)
        n += 1
    # JPEG XT with -n: the residual codestream brings its height in a DNL marker as well, and the reference compares the two
    # frames' dimensions right behind the residual frame header (codestream/image.cpp:1289-1299): refused
    for k, extra in enumerate((["-n"], ["-n", "-z", "2", "-s", "1x1,2x2,2x2"])):
        data = O.reference_encode_hdr(synth.synth_hdr(64, 48, 5 + k) * 4.0, ["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c", "-r12"] + extra)
        px, err = O.reference_decode_status(data)
        assert px is None and err == -1038, err
        with open(os.path.join(OUT, f"dnl_xt_{k}.jpg"), "wb") as f:
            f.write(data)
        manifest[f"dnl_xt_{k}"] = dict(kind="xt", error=err, jpeg_sha256=sha(data))
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print(len(manifest), "cases")


if __name__ == "__main__":
    main()
