#!/usr/bin/env python3
"""Differential campaign on damaged streams: oracle (oracle/jpeg_oracle.c) against the real reference binary
(oracle/_ref/jpeg, build container only), and optionally the product's host decoder against the oracle.

  python tools/damage_campaign.py --per-file 200 [--files a.jpg b.jpg] [--seed 1] [--where any|entropy|header]
                                  [--product]      # libjpeg_amd host entropy decoder vs the oracle (coefficients)

Prints one line per disagreement and a summary; exit code 1 if anything disagreed.
"""
import argparse
import glob
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def one(args):
    path, kind, blob, mode = args
    from oracle import oracle as O
    opx, oerr, owarn = O.decode_status(blob)
    if oerr is None:
        return path, kind, "skip", None
    if mode == "reference":
        rpx, rerr = O.reference_decode_status(blob)
        if rerr == "timeout":  # the reference itself loops forever on some damaged streams: nothing to compare with
            return path, kind, "ref-hangs", None
        if rerr == 0 and oerr == 0:
            if rpx.shape != opx.shape:
                return path, kind, "shape", (rpx.shape, opx.shape)
            nd = int(np.count_nonzero(rpx != opx))
            return path, kind, ("ok" if nd == 0 else "pixels"), nd
        if rerr != 0 and oerr != 0:
            return path, kind, ("ok" if rerr == oerr else "code"), (rerr, oerr)
        return path, kind, "decode-vs-error", (rerr, oerr)
    if mode == "product":
        return (path, kind) + product_vs_oracle(blob)
    raise ValueError(mode)


def product_vs_oracle(blob):
    import damage
    return damage.product_vs_oracle(blob)


def main():
    import damage
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", nargs="*")
    ap.add_argument("--per-file", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--where", default="any")
    ap.add_argument("--jobs", type=int, default=os.cpu_count())
    ap.add_argument("--save", default="")
    ap.add_argument("--product", action="store_true", help="libjpeg_amd host decoder against the oracle instead of oracle against reference")
    a = ap.parse_args()
    files = a.files or sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.jpg")))
    # JPEG XT files and frames whose height arrives in a DNL marker are outside the damaged-stream contract (DESIGN)
    files = [f for f in files if "xt_" not in os.path.basename(f) and "dnl" not in os.path.basename(f)]
    work = []
    for fi, f in enumerate(files):
        data = open(f, "rb").read()
        for kind, blob in damage.cases(data, a.per_file, a.seed * 1000 + fi, a.where):
            work.append((os.path.basename(f), kind, blob, "product" if a.product else "reference"))
    stats = {}
    bad = 0
    with ProcessPoolExecutor(a.jobs) as ex:
        for (path, kind, verdict, detail), w in zip(ex.map(one, work, chunksize=8), work):
            stats[verdict] = stats.get(verdict, 0) + 1
            if verdict not in ("ok", "skip", "ref-hangs"):
                bad += 1
                print(f"{verdict:16s} {path:36s} {kind:16s} {detail}")
                if a.save:
                    os.makedirs(a.save, exist_ok=True)
                    with open(os.path.join(a.save, f"{bad:04d}_{verdict}_{kind}_{path}"), "wb") as fo:
                        fo.write(w[2])
    print("summary:", stats, "of", len(work))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
