#!/usr/bin/env python3
"""CPU-only differential campaign: N seeded random VALID streams (every sampling layout Pillow writes, grey, progressive,
optimised tables, restart intervals, noise and saturated content, one in five turned into a 12-bit stream by synth.to_12bit)
through the product's host decoder against the oracle, coefficient by coefficient (tests/damage.product_vs_oracle).
The damaged counterpart is tools/damage_campaign.py --product; the GPU campaigns are tools/random_campaign.py.
    python tools/host_campaign.py 4000 [jobs]"""
import collections
import os
import sys
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import damage  # noqa: E402
from libjpeg_amd import synth  # noqa: E402


def one(t):
    rng = np.random.default_rng(900000 + t)
    w, h = int(rng.integers(1, 500)), int(rng.integers(1, 400))
    sub = ["444", "422", "420", "gray"][int(rng.integers(0, 4))]
    q = int(rng.choice([3, 20, 50, 75, 85, 95, 100]))
    dri = int(rng.choice([0, 0, 1, 3, 8, 40]))
    prog, opt = bool(rng.integers(0, 4) == 0), bool(rng.integers(0, 2))
    img = synth.synth_image(w, h, 5000 + t, channels=1 if sub == "gray" else 3)
    style = int(rng.integers(0, 4))
    if style == 0:
        img = rng.integers(0, 256, img.shape).astype(np.uint8)
    elif style == 1:
        img = (img > 128).astype(np.uint8) * 255
    try:
        data = synth.encode_jpeg(img, q, sub if sub != "gray" else "444", restart_mcus=dri, optimize=opt, progressive=prog)
    except OSError:
        return "refused by the test encoder", None
    if rng.integers(0, 5) == 0 and not prog:
        data = synth.to_12bit(data, int(rng.choice([1, 7, 16])))
    verdict, detail = damage.product_vs_oracle(data)
    return verdict, (t, w, h, sub, q, dri, prog, opt, detail) if verdict != "ok" else None


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    jobs = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
    stats, bad = collections.Counter(), 0
    with ProcessPoolExecutor(jobs) as ex:
        for verdict, detail in ex.map(one, range(n), chunksize=16):
            stats[verdict] += 1
            if detail is not None and verdict not in ("skip",):
                bad += 1
                print(verdict, detail, flush=True)
    print(dict(stats))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
