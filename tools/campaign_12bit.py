#!/usr/bin/env python3
"""One-off differential campaign for the 12-bit kernels (round 6: the colour stage comes in two flavours, colour12 in kernels.hip): N seeded
random 12-bit streams (synth.to_12bit: an 8-bit stream's entropy coded data under deltas times `scale`) of random size, layout, quality,
content and scale through the decoder object against the oracle; the scales and contents are chosen so that frames fall on both sides of
the one-sum gate (narrow12_colour) and of the fused kernels' own gates.  Prints the kernels met; exit code 1 on any difference."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from libjpeg_amd import api, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

N = int(os.environ.get("N", "600"))
rng = np.random.default_rng(int(os.environ.get("SEED", "1212")))
d = api.Decoder(0)
kernels, bad, skipped = collections.Counter(), 0, 0
for t in range(N):
    big = rng.integers(0, 10) == 0
    w, h = (int(rng.integers(600, 2200)), int(rng.integers(400, 1300))) if big else (int(rng.integers(1, 500)), int(rng.integers(1, 400)))
    sub = ["444", "422", "420", "gray"][int(rng.integers(0, 4))]
    q = int(rng.choice([20, 50, 75, 85, 95, 100]))
    dri = int(rng.choice([0, 1, 4, 8]))
    scale = int(rng.choice([3, 9, 16, 16, 24, 40]))
    img = synth.synth_image(w, h, 7000 + t, channels=1 if sub == "gray" else 3)
    style = int(rng.integers(0, 4))
    if style == 0:
        img = rng.integers(0, 256, img.shape).astype(np.uint8)
    elif style == 1:  # saturated graphics: large chroma ranges
        img = (img > 128).astype(np.uint8) * 255
    try:
        data = synth.to_12bit(synth.encode_jpeg(img, q, sub if sub != "gray" else "444", restart_mcus=dri), scale)
    except (OSError, ValueError):
        skipped += 1
        continue
    try:
        exp = O.decode16(data)
    except Exception:  # deltas beyond what the frame header may carry
        skipped += 1
        continue
    for mode in ("host", "auto"):
        f = d.read(data, entropy=mode)
        got = d.reconstruct()
        kernels[api.kernel_name(f)] += 1
        if got.shape != exp.shape or not np.array_equal(got, exp):
            bad += 1
            print("DIFFERENCE", t, w, h, sub, q, dri, scale, style, mode, api.kernel_name(f), list(f.range_max)[:3], flush=True)
print(f"{N} streams, seed {os.environ.get('SEED', '1212')}: {bad} differences, {skipped} skipped")
for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]):
    print(f"  {v:5d}  {k}")
sys.exit(1 if bad else 0)
