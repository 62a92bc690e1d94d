#!/usr/bin/env python3
"""CPU only: every JPEG XT golden under the seeded corruptions of tests/damage.py (and random bytes in front of the first scan) -- the product's
HOST decoder against the oracle: verdicts, and where both decode, the coefficient planes of both frames as the merge sees them
(visible scans moved up by the hidden bits, hidden refinement scans applied or not).  What tools/xt_gpu_damage_campaign.py checks through
the pixels on a GPU box, one level down and without a device.    SEED=1 PER_FILE=12 [JOBS=6] python tools/xt_host_damage_campaign.py"""
import collections
import glob
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import damage  # noqa: E402

SEED = int(os.environ.get("SEED", "1"))
PER = int(os.environ.get("PER_FILE", "12"))


def files():
    g = os.path.join(ROOT, "tests", "golden")
    return sorted(glob.glob(os.path.join(g, "xt_*.jpg"))) + sorted(glob.glob(os.path.join(g, "xt_*", "*.jpg")))


def one(job):
    fi, path = job
    from libjpeg_amd import api
    from oracle import oracle as O
    data = open(path, "rb").read()
    name = os.path.basename(path)[:-4]
    rng = np.random.default_rng(SEED * 100000 + fi)
    hdr = damage.entropy_start(data)
    cases = []
    for k in range(PER):
        b = bytearray(data)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(2, hdr))] = int(rng.integers(0, 256))
        cases.append(("hdr%d" % k, bytes(b)))
    for where in ("any", "entropy"):
        cases += list(damage.cases(data, PER, SEED * 100000 + fi + (500 if where == "entropy" else 0), where))
    count = collections.Counter()
    bad = []
    d = api.Decoder(None)
    for kind, blob in cases:
        try:
            oerr = O.decode_xt_status(blob)[2]
        except Exception:  # noqa: BLE001
            oerr = None
        if oerr is None:
            count["outside the restatement"] += 1
            continue
        if oerr == 0:  # (the alpha channel's codestreams fail the read like the image's own)
            aerr = O.alpha_read_error(blob)
            if aerr not in (None, 0):
                oerr = aerr
        try:
            f = d.read(blob)
            perr = 0
        except api.MijpegError as e:
            perr = e.code
        if perr in (-1034, -1042):
            count["declined"] += 1
            continue
        if perr != oerr:
            count["MISMATCH code"] += 1
            bad.append((name, kind, oerr, perr))
            continue
        if oerr:
            count["same error"] += 1
            continue
        try:
            _, pl = O.decode_xt_planes(blob)
            _, rp = O.decode_xt_residual_planes(blob)
        except Exception:  # noqa: BLE001 -- (a picture without a residual frame, the L chain alone: the planes of the plain restatement are not these)
            count["same verdict (no planes to compare)"] += 1
            continue
        def same(got, want):
            a, w = np.asarray(got).reshape(-1, 64).astype(np.int64), np.asarray(want).reshape(-1, 64).astype(np.int64)
            if np.array_equal(a, w):
                return True
            # a component that appears in no scan: the oracle keeps zeros and remembers, the product stores its stand-in -- a DC that
            # cancels the level shift under a quantiser table of ones (HostDecoder::fill_unseen_components): the same samples
            return not w.any() and not a[:, 1:].any() and len(set(a[:, 0].tolist())) == 1
        ok = all(same(d.coefficients(c), pl[c]) for c in range(f.components))
        x = d.xt_params()
        if ok and not x.no_residual and x.residual.components:
            ok = all(same(d.residual_coefficients(c), rp[c]) for c in range(len(rp)))
        if ok:
            count["same planes"] += 1
        else:
            count["MISMATCH planes"] += 1
            bad.append((name, kind, "planes"))
    d.close()
    return count, bad


def main():
    total = collections.Counter()
    bad = []
    with ProcessPoolExecutor(int(os.environ.get("JOBS", "6"))) as ex:
        for c, b in ex.map(one, list(enumerate(files()))):
            total.update(c)
            bad += b
    print("seed", SEED, dict(total))
    for b in bad[:40]:
        print("MISMATCH", b)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
