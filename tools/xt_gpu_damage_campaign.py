#!/usr/bin/env python3
"""GPU box: the stream classes of round 5 (tests/golden/xt_lonly, xt_alpha, xt_lossless) under the seeded corruptions of tests/damage.py and
random bytes in front of the first scan -- where oracle and product both decode, the product's picture (and alpha plane) on the device must
be the oracle's.  (Return codes three-way with the reference binary: tools/box_campaign.py, CPU.)   SEED=1 PER_FILE=12 [ENTROPY=prefer-gpu] [FILES=all|every] python tools/xt_gpu_damage_campaign.py"""
import collections
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import damage  # noqa: E402
from libjpeg_amd import api  # noqa: E402
from oracle import oracle as O  # noqa: E402

SEED = int(os.environ.get("SEED", "1"))
PER = int(os.environ.get("PER_FILE", "12"))


def cases():
    g = os.path.join(ROOT, "tests", "golden")
    files = sorted(glob.glob(os.path.join(g, "xt_lonly", "*.jpg")))[::2] + sorted(glob.glob(os.path.join(g, "xt_alpha", "*.jpg"))) + \
        sorted(glob.glob(os.path.join(g, "xt_lossless", "*.jpg"))) + sorted(glob.glob(os.path.join(g, "xt_grey", "*.jpg"))) + sorted(glob.glob(os.path.join(g, "xt_int8", "*.jpg")))[:10]
    if os.environ.get("FILES") == "every":  # every class of tests/golden/xt_*/ (general transformations, integer output, boxes, ...) and the files beside them
        files = sorted(glob.glob(os.path.join(g, "xt_*.jpg"))) + sorted(glob.glob(os.path.join(g, "xt_*", "*.jpg")))
    if os.environ.get("FILES") == "all":  # every JPEG XT golden, the ones with restart markers included (what the device entropy decoders take)
        files = sorted(glob.glob(os.path.join(g, "xt_*.jpg"))) + files
    if os.environ.get("CLASSES") == "refinement":
        # frames whose scans run as pipeline windows on the host -- profile C with hidden bits in either frame, progressive pictures:
        # the refinement chains on bit masks, their appliers, the residual rows that go to the device early
        files = sorted(f for f in glob.glob(os.path.join(g, "*.jpg")) if "prog" in os.path.basename(f) or
                       any(p.startswith(("R", "rR")) and p[-1].isdigit() for p in os.path.basename(f)[:-4].split("_")))
    for fi, f in enumerate(files):
        data = open(f, "rb").read()
        name = os.path.basename(f)[:-4]
        rng = np.random.default_rng(SEED * 100000 + fi)
        hdr = damage.entropy_start(data)
        for k in range(PER):
            b = bytearray(data)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(2, hdr))] = int(rng.integers(0, 256))
            yield name, "hdr%d" % k, bytes(b)
        for where in ("any", "entropy"):
            for kind, blob in damage.cases(data, PER, SEED * 100000 + fi + (500 if where == "entropy" else 0), where):
                yield name, kind, blob


def reference_crashes(blob):
    import subprocess
    import tempfile
    if not O.have_reference():
        return False
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
        src, dst = os.path.join(d, "i.jpg"), os.path.join(d, "o.out")
        open(src, "wb").write(blob)
        try:
            args = [O.REF_BIN] + (["-al", os.path.join(d, "a.out")] if b"ALFA" in blob else []) + [src, dst]  # (the alpha plane is asked for, too)
            return subprocess.run(args, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=20).returncode < 0
        except subprocess.TimeoutExpired:
            return True


def main():
    dec = api.Decoder(0)
    count = collections.Counter()
    bad = []
    for name, kind, blob in cases():
        try:
            want, is_float, oerr = O.decode_xt_status(blob)
        except Exception:  # noqa: BLE001
            want, oerr = None, None
        if oerr is None:  # no XT file any more, or outside the restatement: the plain oracle
            try:
                want, oerr, _ = O.decode_status(blob)
            except Exception:  # noqa: BLE001
                want, oerr = None, None
        try:
            dec.read(blob, entropy=os.environ.get("ENTROPY", "auto"))  # ("prefer-gpu": the device entropy decoders wherever they qualify, however small the file)
            got = dec.reconstruct()
            perr = 0
        except api.MijpegError as e:
            got, perr = None, e.code
        if oerr is None or perr in (-1034, -1042):
            count["declined / outside the restatement"] += 1
            continue
        aerr = O.alpha_read_error(blob) if oerr == 0 else None
        if oerr == 0 and aerr not in (None, 0):
            oerr, want = aerr, None  # (the alpha channel's codestreams fail the read)
        if oerr != perr:
            count["MISMATCH code"] += 1; bad.append((name, kind, oerr, perr)); continue
        if oerr != 0:
            count["same error"] += 1
            continue
        if not np.array_equal(np.asarray(got).reshape(-1).astype(np.uint16), np.asarray(want).reshape(-1).astype(np.uint16)):
            if reference_crashes(blob):  # (undefined behaviour in the reference: the CPU campaigns skip these streams, nothing to be equal to)
                count["differ where the reference binary crashes"] += 1
                continue
            count["MISMATCH pixels"] += 1; bad.append((name, kind, "pixels")); continue
        acodes, _, _, _, _, ae = O.decode_alpha(blob)
        if acodes is not None and ae == 0:
            a = dec.alpha_channel()
            if a is None:
                count["MISMATCH alpha missing"] += 1; bad.append((name, kind, "alpha missing")); continue
            if not np.array_equal(np.asarray(a.reconstruct()).reshape(-1).astype(np.uint16), acodes.reshape(-1)):
                if reference_crashes(blob):
                    count["differ where the reference binary crashes"] += 1
                    continue
                count["MISMATCH alpha pixels"] += 1; bad.append((name, kind, "alpha pixels")); continue
            count["same picture and alpha plane"] += 1
        else:
            count["same picture"] += 1
    print("seed", SEED, dict(count))
    for b in bad[:40]:
        print("MISMATCH", b)


if __name__ == "__main__":
    main()
