#!/usr/bin/env python3
"""tests/golden/stop_counts.json: how many times the REAL reference library returns from JPEG::Read under JPGFLAG_DECODER_STOP_ROW,
_MCU and _SCAN | _ROW until the image is read (interface/jpeg.cpp:244-354), per golden stream -- tests/cxx/marker_calls.cpp linked
against the reference's own objects (oracle/_ref/obj, `make -C oracle ref`).  Run in the build container."""
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.environ.get("LIBJPEG_REFERENCE", "/root/reference")
CASES = ["pil_200x120_420_dri8", "ref_75x45_420_dri2", "pil_70x40_gray", "pilprog_75x45_420", "refprog_64x64_444_dri5", "xt_64x48_444", "p12_64x48_444",
         "ref_97x61_3x3", "pilprog_200x130_422", "xt_200x120_420_R3_rR4", "xt_129x71_420_R2_rR3_dri3", "xt_64x48_444_R4", "pil_90x60_cmyk",
         "ref_23x50_lumasub", "ref_97x61_mixed", "ref_97x61_411", "refc_83x47_440",
         # alpha channels: the scans of the alpha image's codestreams count behind the image's own (round 6)
         "xt_alpha/a8_beside_hdr", "xt_alpha/a16_residual", "xt_alpha/a8_residual_hidden", "xt_alpha/af_beside_hdr", "xt_alpha/a8_420",
         # frames the library walks sequentially: a height that arrives in a DNL marker, the residual scan types of part 8
         "dnl/dnl_1x2_33_prog", "dnl/dnl_420_32_dri2", "dnl/dnl_gray_33", "dnl/dnl_1x2_16_bl", "xt_lossless/rgb8_ro_dri4", "xt_lossless/rgb8_ro_rv", "xt_lossless/ghdr_ro"]


def main():
    objs = [o for o in glob.glob(os.path.join(ROOT, "oracle", "_ref", "obj", "**", "*.o"), recursive=True) if os.sep + "cmd" + os.sep not in o]
    if not objs:
        sys.exit("oracle/_ref/obj is missing: run `make -C oracle ref` first")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "ref")
        subprocess.run(["g++", "-O1", "-w", "-DUSE_AUTOCONF", "-fno-exceptions", "-I", REF, "-I", os.path.join(ROOT, "oracle", "_ref", "gen"),
                        os.path.join(ROOT, "tests", "cxx", "marker_calls.cpp"), *objs, "-o", exe, "-lm"], check=True)
        out = {}
        for name in CASES:
            out[name] = {}
            for mode in ("row", "mcu", "scanrow"):
                r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", name + ".jpg"), mode], capture_output=True, text=True, check=True)
                out[name][mode] = r.stdout.strip().splitlines()
    with open(os.path.join(ROOT, "tests", "golden", "stop_counts.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(len(out), "streams")


if __name__ == "__main__":
    main()
