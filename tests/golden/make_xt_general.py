#!/usr/bin/env python3
"""Regenerate tests/golden/xt_general/: JPEG XT profile C streams beyond the reference encoder's default output, and what the
REAL reference decoder makes of each (float32 pixels as its CLI writes them into a PFM, or the JPGERR_* code it fails with).

  * encoder variants: `-xyz` / `-cxyz` (free-form L / R / C transformations in MTRX boxes), an 8-bit residual (no -r12);
  * hand-made variants (tests/xt_craft.py): parametric curves (CURV boxes) as L, Q and R2 tables, the residual DCT bypass with
    and without noise shaping, and merging specifications the reference rejects.

Run in the build container (needs oracle/_ref/jpeg):   python tests/golden/make_xt_general.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import xt_craft  # noqa: E402
from libjpeg_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "xt_general")
BASE = ["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c"]


def main():
    if not O.have_reference():
        sys.exit("oracle/_ref/jpeg is missing: run `make -C oracle ref` first")
    os.makedirs(OUT, exist_ok=True)
    cases = {}
    hdr = synth.synth_hdr(48, 32, 5) * 4.0
    cases["enc_xyz"] = O.reference_encode_hdr(hdr, BASE + ["-r12", "-xyz"])
    cases["enc_cxyz"] = O.reference_encode_hdr(hdr, BASE + ["-r12", "-cxyz"])
    cases["enc_residual8"] = O.reference_encode_hdr(hdr, BASE)
    cases["enc_xyz_420_R1"] = O.reference_encode_hdr(synth.synth_hdr(75, 45, 6) * 4.0, BASE + ["-r12", "-xyz", "-s", "1x1,2x2,2x2", "-R", "1"])
    a = O.reference_encode_hdr(hdr, BASE + ["-r12"])
    for k, v in xt_craft.variants(a).items():
        cases["a_" + k] = v
    b = O.reference_encode_hdr(synth.synth_hdr(75, 45, 7) * 4.0, BASE + ["-r12", "-s", "1x1,2x2,2x2", "-R", "2", "-rR", "3"])
    vb = xt_craft.variants(b)
    for k in ("q_and_r2", "bypass_noise", "l_gamma_curve", "r2_gamma"):
        cases["b_" + k] = vb[k]
    manifest = {}
    for name, blob in cases.items():
        with open(os.path.join(OUT, name + ".jpg"), "wb") as f:
            f.write(blob)
        ent = {"jpeg_sha256": hashlib.sha256(blob).hexdigest()}
        px, err = O.reference_decode_status(blob)  # the verdict only (its reader expects PNM): an int < 0 is the reference's error code
        if not (isinstance(err, int) and err < 0):
            err = 0
        binfile = os.path.join(OUT, name + ".bin")
        if err == 0:
            ref = O.reference_decode_hdr(blob).astype("<f4")
            ent.update(error=0, height=int(ref.shape[0]), width=int(ref.shape[1]), pixels_sha256=hashlib.sha256(ref.tobytes()).hexdigest())
            with open(binfile, "wb") as f:
                f.write(ref.tobytes())
        else:
            ent.update(error=err)
            if os.path.exists(binfile):
                os.remove(binfile)
        manifest[name] = ent
        print(f"{name:32s} reference: {'picture' if err == 0 else err}")
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
