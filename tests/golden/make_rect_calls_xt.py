#!/usr/bin/env python3
"""Writes tests/golden/rect_calls_xt/: request SEQUENCES against JPEG XT (profile C) images and what the REAL reference library
left in the client's bitmaps (oracle/_ref/rect_calls_ref = tests/cxx/rect_calls.cpp linked against the reference's objects).
The residual image has row cursors and upsamplers of its own beside the legacy image's (control/blockbitmaprequester.cpp:
228-232, 356-372, 1118-1146, 1197-1222), so a request shows what the calls before it left of BOTH.
Streams (reference encoder): 4:4:4 and 4:2:0 legacy frames with a 12-bit 4:4:4 residual, a 4:2:0 residual (-sr), hidden bits in both
frames (-R / -rR), the 8-bit integer flavour; sequences: all three components with upsampling and colour transformation (what
the command line asks for) as skipped / unaligned / windowed stripes and low bitmaps.
    python tests/golden/make_rect_calls_xt.py          (build container: needs oracle/_ref/rect_calls_ref and oracle/_ref/jpeg)"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from libjpeg_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(HERE, "rect_calls_xt")
W, H = 59, 45
HDR = ["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c", "-r12"]


def streams():
    hdr = synth.synth_hdr(W, H, 5) * 4.0
    img = synth.synth_image(W, H, 6)
    yield "x444", O.reference_encode_hdr(hdr, HDR)
    yield "x420", O.reference_encode_hdr(hdr, HDR + ["-s", "1x1,2x2,2x2"])
    yield "x420_sr420", O.reference_encode_hdr(hdr, HDR + ["-s", "1x1,2x2,2x2", "-sr", "1x1,2x2,2x2"])
    yield "x422_hidden", O.reference_encode_hdr(hdr, HDR + ["-s", "1x1,2x1,2x1", "-rR", "2", "-R", "1"])
    yield "i420", O.reference_encode(img, ["-r", "-q", "85", "-Q", "90", "-s", "1x1,2x2,2x2"])


def scripts():
    stripes = [(0, y, -1, min(y + 7, H - 1), 0, 2, 1, 1, 8) for y in range(0, H, 8)]
    yield "stripes", stripes
    yield "skip_two", stripes[2:]
    yield "every_other", stripes[::2]
    yield "unaligned", [(0, 3, -1, 20, 0, 2, 1, 1, 0), (0, 21, -1, 21, 0, 2, 1, 1, 0), (0, 22, -1, 29, 0, 2, 1, 1, 0)]  # (six block rows in all: the cursors of a 4:4:4 image must not leave it)
    yield "window_then_rest", [(13, 5, 50, 22, 0, 2, 1, 1, 0), (0, 24, -1, -1, 0, 2, 1, 1, 0)]
    yield "low_bitmaps", [(0, 0, -1, 31, 0, 2, 1, 1, 8), (0, 8, -1, 31, 0, 2, 1, 1, 16), (0, 32, -1, -1, 0, 2, 1, 1, 0)]


def main():
    if not os.path.exists(O.REF_RECT_CALLS) or not O.have_reference():
        raise SystemExit("oracle/_ref/rect_calls_ref or oracle/_ref/jpeg is missing: make -C oracle ref")
    os.makedirs(OUT, exist_ok=True)
    manifest = {}
    for sname, data in streams():
        assert data, sname
        with open(os.path.join(OUT, sname + ".jpg"), "wb") as f:
            f.write(data)
        for qname, requests in scripts():
            name = f"{sname}__{qname}"
            lines, planes = O.run_requests_client(O.REF_RECT_CALLS, data, requests)
            assert planes is not None, (name, lines)
            planes.tofile(os.path.join(OUT, name + ".bin"))
            manifest[name] = dict(stream=sname, requests=[list(r) for r in requests], shape=list(planes.shape), dtype=str(planes.dtype),
                                  calls=[ln.split()[2:] for ln in lines[1:]])
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)
    print(len(manifest), "request sequences on JPEG XT streams written by the reference")


if __name__ == "__main__":
    main()
