#!/usr/bin/env python3
"""Writes tests/golden/rect_calls/: request SEQUENCES against one decoded image and what the REAL reference library left
in the client's bitmaps (oracle/_ref/rect_calls_ref = tests/cxx/rect_calls.cpp linked against the reference's objects).
Streams: hand-written frames of two and four components with mixed sampling (tests/craft.py), the committed CMYK fixture with
one sampling byte changed, three-component frames; sequences: the command line's own loops (cmd/reconstruct.cpp:272-342) and
orders no sane client uses (repeated / skipped / unaligned stripes, component subsets under a colour transformation).
    python tests/golden/make_rect_calls.py          (build container: needs oracle/_ref/rect_calls_ref)"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import craft  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(HERE, "rect_calls")


def pgx(w, h, nc, order=None):
    return [(0, y, -1, min(y + 7, h - 1), c, c, 1, 1, 8) for c in (order or range(nc)) for y in range(0, h, 8)]


def ppm(w, h, nc):
    return [(0, y, -1, min(y + 7, h - 1), 0, nc - 1, 1, 1, 8) for y in range(0, h, 8)]


def cases():
    mixed = {"c2_11_21": [(1, 1), (2, 1)], "c2_11_22": [(1, 1), (2, 2)], "c4_11_11_31_11": [(1, 1), (1, 1), (3, 1), (1, 1)],
             "c4_22_11_11_22": [(2, 2), (1, 1), (1, 1), (2, 2)], "c3_21_12_11": [(2, 1), (1, 2), (1, 1)], "c2_14_41": [(1, 4), (4, 1)]}
    for k, (name, samp) in enumerate(mixed.items()):
        rng = np.random.default_rng(300 + k)
        w, h = 61 + 7 * k, 45 + 5 * k
        data = craft.craft_stream(rng, samp, w, h, dri=[0, 2, 0, 5, 1, 0][k])
        nc = len(samp)
        yield name + "_pgx", data, pgx(w, h, nc)
        yield name + "_pgx_backwards", data, pgx(w, h, nc, list(reversed(range(nc))))
        yield name + "_stripes", data, ppm(w, h, nc)
        yield name + "_whole_twice", data, [(0, 0, -1, -1, 0, nc - 1, 1, 1, 0)] * 2
    with open(os.path.join(HERE, "pil_90x60_cmyk.jpg"), "rb") as f:
        cmyk = bytearray(f.read())
    assert cmyk[104] == 0x11
    cmyk[104] = 0x21  # third component 2x1: the others become subsampled, it stays the unsubsampled one
    yield "cmyk_byte104_pgx", bytes(cmyk), pgx(90, 60, 4)
    with open(os.path.join(HERE, "pil_200x120_420_dri8.jpg"), "rb") as f:
        c420 = f.read()
    w, h = 200, 120
    yield "c420_unaligned", c420, [(0, 3, -1, 20, 0, 2, 1, 1, 0), (0, 21, -1, 21, 0, 2, 1, 1, 0), (0, 22, -1, 77, 0, 2, 1, 1, 0), (0, 78, -1, -1, 0, 2, 1, 1, 0)]
    yield "c420_subsets_under_ycc", c420, [(0, 0, -1, 39, 0, 2, 1, 1, 0), (0, 40, -1, 79, 1, 1, 1, 1, 0), (0, 80, -1, -1, 0, 1, 1, 1, 0)]
    yield "c420_first_call_decides_the_transformer", c420, [(0, 0, -1, 39, 0, 2, 1, 0, 0), (0, 40, -1, -1, 0, 2, 1, 1, 0)]
    yield "c420_repeat_and_skip", c420, [(0, 0, -1, 15, 0, 2, 1, 1, 0), (0, 0, -1, 15, 0, 2, 1, 1, 0), (0, 48, -1, 63, 0, 2, 1, 1, 0), (0, 16, -1, 31, 0, 2, 1, 1, 0)]
    yield "c420_low_bitmaps", c420, [(0, 0, -1, 31, 0, 2, 1, 1, 8), (0, 8, -1, 31, 0, 2, 1, 1, 5), (0, 8, -1, 31, 0, 2, 1, 1, 16), (0, 32, -1, -1, 0, 2, 1, 1, 0)]
    yield "c420_noup_then_up", c420, [(0, 0, -1, 31, 1, 1, 0, 0, 0), (0, 0, -1, 31, 0, 2, 1, 1, 0), (0, 32, -1, 63, 2, 2, 0, 0, 0), (0, 32, -1, -1, 0, 2, 1, 1, 0)]
    with open(os.path.join(HERE, "pil_80x48_444.jpg"), "rb") as f:
        c444 = f.read()
    yield "c444_subset_then_all", c444, [(0, 0, -1, 15, 1, 1, 1, 1, 0), (0, 0, -1, 15, 0, 2, 1, 1, 0), (0, 16, -1, -1, 0, 2, 1, 1, 0)]
    yield "c444_window", c444, [(13, 5, 50, 30, 0, 2, 1, 1, 0), (0, 31, -1, -1, 0, 2, 1, 1, 0)]


def main():
    if not os.path.exists(O.REF_RECT_CALLS):
        raise SystemExit("oracle/_ref/rect_calls_ref is missing: make -C oracle ref")
    os.makedirs(OUT, exist_ok=True)
    manifest = {}
    for name, data, requests in cases():
        lines, planes = O.run_requests_client(O.REF_RECT_CALLS, data, requests)
        assert planes is not None, (name, lines)
        with open(os.path.join(OUT, name + ".jpg"), "wb") as f:
            f.write(data)
        planes.tofile(os.path.join(OUT, name + ".bin"))
        manifest[name] = dict(requests=[list(r) for r in requests], shape=list(planes.shape), dtype=str(planes.dtype),
                              calls=[ln.split()[2:] for ln in lines[1:]])
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0, sort_keys=True)
    print(len(manifest), "request sequences written by the reference")


if __name__ == "__main__":
    main()
