#!/usr/bin/env python3
"""Regenerate tests/golden/xt_int8/: JPEG XT profile C streams with an INTEGER output of eight bits -- what the reference encoder
writes for `jpeg -r -q .. -Q .. in.ppm` (an 8-bit legacy codestream, a residual codestream, an output conversion without extra
range bits: boxes/outputconversionbox.hpp, colortrafo/colortransformerfactory.cpp:300-372) -- and what the REAL reference decoder
makes of each: its PPM samples, or the JPGERR_* code it fails with.

  * encoder variants: 4:4:4 / 4:2:0 / 4:2:2, a 12-bit residual (-r12), hidden residual bits (-rR), the optimiser (-v / -N runs),
    RGB legacy (-c), free-form transformations (-xyz);
  * hand-made variants (tests/xt_craft.py): parametric Q / R2 / L curves scaled to the 8-bit output, the residual DCT bypass.

Run in the build container (needs oracle/_ref/jpeg):   python tests/golden/make_xt_int8.py
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import xt_craft  # noqa: E402
from libjpeg_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "xt_int8")
BASE = ["-r", "-q", "85", "-Q", "90"]


def main():
    if not O.have_reference():
        sys.exit("oracle/_ref/jpeg is missing: run `make -C oracle ref` first")
    os.makedirs(OUT, exist_ok=True)
    a_img, b_img = synth.synth_image(48, 32, 5), synth.synth_image(83, 45, 7)
    cases = {
        "enc_444": O.reference_encode(a_img, BASE),
        "enc_420": O.reference_encode(b_img, BASE + ["-s", "1x1,2x2,2x2"]),
        "enc_422_coarse": O.reference_encode(b_img, ["-r", "-q", "30", "-Q", "40", "-s", "1x1,2x1,2x1"]),
        "enc_r12": O.reference_encode(b_img, ["-r", "-q", "60", "-Q", "70", "-r12"]),
        "enc_rR2": O.reference_encode(b_img, BASE + ["-rR", "2"]),  # (the encoder refuses hidden residual bits with subsampling)
        "enc_v": O.reference_encode(a_img, BASE + ["-v"]),
        "enc_N": O.reference_encode(a_img, BASE + ["-N"]),
        "enc_c": O.reference_encode(a_img, BASE + ["-c"]),
        "enc_xyz": O.reference_encode(a_img, BASE + ["-xyz"]),
    }
    assert all(cases.values()), [k for k, v in cases.items() if not v]  # (an encoder error leaves an empty file)
    for k, v in xt_craft.variants(cases["enc_444"]).items():
        cases["a_" + k] = v
    vb = xt_craft.variants(cases["enc_rR2"])
    for k in ("q_and_r2", "bypass_noise", "l_gamma_curve", "r2_gamma"):
        cases["b_" + k] = vb[k]
    manifest = {}
    for name, blob in cases.items():
        with open(os.path.join(OUT, name + ".jpg"), "wb") as f:
            f.write(blob)
        ent = {"jpeg_sha256": hashlib.sha256(blob).hexdigest()}
        px, err = O.reference_decode_status(blob)
        if not (isinstance(err, int) and err < 0):
            err = 0
        binfile = os.path.join(OUT, name + ".bin")
        if err == 0:
            assert px.dtype == "uint8" and px.ndim == 3, (name, px.dtype, px.shape)
            ent.update(error=0, height=int(px.shape[0]), width=int(px.shape[1]), pixels_sha256=hashlib.sha256(px.tobytes()).hexdigest())
            with open(binfile, "wb") as f:
                f.write(px.tobytes())
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
