#!/usr/bin/env python3
"""CPU only, build container (needs oracle/_ref/jpeg): damage in the part of a file that tests/damage.py's uniform positions
rarely hit -- the boxes, tables and, in JPEG XT files, the residual codestream in front of the legacy frame's first scan --
plus the usual seeded corruptions, on the stream classes of round 4 (one-component and integer JPEG XT, `-c` files) and the
older goldens.  Three-way: the reference binary (return code, pixels), the oracle (XT restatement, or the plain one where the
file is no XT file any more), the product's host decoder (return code; -1034 / -1042 = declined, never a wrong picture).

    python tools/box_campaign.py xt SEED      # JPEG XT goldens (tests/golden/xt_*)
    python tools/box_campaign.py plain SEED   # 8- and 12-bit goldens incl. the merging specifications without residual

Prints a counter and every case that is not `ok`; `--keep DIR` stores those streams.  What it found in round 4 is pinned in
tests/test_xt_boxes.py."""
import collections
import ctypes as C
import glob
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import damage  # noqa: E402
from conftest import P12_CASES, SMALL_CASES, golden_jpeg  # noqa: E402
from libjpeg_amd import api  # noqa: E402
from oracle import oracle as O  # noqa: E402


DECLINED = (-1034, -1042)  # NOT_IMPLEMENTED / a Huffman table beyond the accelerated path: refusals, never a wrong picture


def reference(data):
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
        src, dst = os.path.join(d, "in.jpg"), os.path.join(d, "out.ppm")
        with open(src, "wb") as f:
            f.write(data)
        try:
            r = subprocess.run([O.REF_BIN, src, dst], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=20)
        except subprocess.TimeoutExpired:
            return None, "timeout"
        if r.returncode < 0:
            return None, "crash"
        m = re.search(rb"failed - error (-?\d+)", r.stderr)
        if m:
            return None, int(m.group(1))
        try:
            with open(dst, "rb") as f:
                magic = f.read(2)
            return (O.read_pfm_reference(dst) if magic in (b"PF", b"Pf") else O.read_pnm_any(dst)), 0
        except Exception:  # noqa: BLE001
            return None, "no output"


def product(blob):
    d = api.Decoder(None)
    try:
        d.read(blob)
        return 0
    except api.MijpegError as e:
        return e.code
    finally:
        d.close()


def plain_oracle(blob):
    """-> (pixels or None, the reference's code or None where the restatement does not follow)"""
    info = O.OjInfo()
    if O.lib().oj_read_info(blob, len(blob), C.byref(info)):
        return None, (info.ref_error or None)
    planes = [np.zeros((info.bh[c], info.bw[c], 64), np.int32) for c in range(info.ncomp)]
    ptrs = (C.c_void_p * 4)(*[p.ctypes.data for p in planes] + [None] * (4 - info.ncomp))
    rc = O.lib().oj_decode_coefficients(blob, len(blob), C.byref(info), ptrs)
    if rc:
        return None, (info.ref_error if (rc != -2 and info.ref_error) else None)
    try:
        return (O.reconstruct16(info, planes) if info.precision > 8 else O.reconstruct(info, planes)), 0
    except ValueError:
        return None, None


def header_cases(data, n, rng):
    hdr = damage.entropy_start(data)
    for k in range(n):
        b = bytearray(data)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(2, hdr))] = int(rng.integers(0, 256))
        yield "hdr%d" % k, bytes(b)
    for k in range(n // 2):  # a byte missing / a byte too many: segment and box lengths stop fitting
        p = int(rng.integers(2, hdr))
        yield "del%d" % k, data[:p] + data[p + 1:]
        p = int(rng.integers(2, hdr))
        yield "ins%d" % k, data[:p] + bytes([int(rng.integers(0, 256))]) + data[p:]


def work_list(mode, seed):
    if mode == "xt":
        g = os.path.join(ROOT, "tests", "golden")
        files = sorted(glob.glob(os.path.join(g, "xt_grey", "*.jpg"))) + sorted(glob.glob(os.path.join(g, "xt_int8", "*.jpg")))[:14] + \
            sorted(glob.glob(os.path.join(g, "xt_int16", "*.jpg"))) + sorted(glob.glob(os.path.join(g, "xt_*x*.jpg")))
    elif mode == "r5":  # the stream classes of round 5: specifications without residual, alpha channels, lossless coding
        g = os.path.join(ROOT, "tests", "golden")
        files = sorted(glob.glob(os.path.join(g, "xt_lonly", "*.jpg")))[seed % 3::3] + sorted(glob.glob(os.path.join(g, "xt_alpha", "*.jpg"))) + \
            sorted(glob.glob(os.path.join(g, "xt_lossless", "*.jpg")))
    if mode in ("xt", "r5"):
        streams = [(os.path.basename(f)[:-4], open(f, "rb").read()) for f in files]
    else:
        streams = [(n, golden_jpeg(n)) for n in SMALL_CASES + P12_CASES]
    for fi, (name, data) in enumerate(streams):
        rng = np.random.default_rng(seed * 100000 + fi)
        try:
            yield from ((name, k, b) for k, b in header_cases(data, 14 if mode in ("xt", "r5") else 10, rng))
        except Exception:  # noqa: BLE001
            continue
        if mode in ("xt", "r5"):
            for where in ("any", "entropy"):
                for kind, blob in damage.cases(data, 10, seed * 100000 + fi + (500 if where == "entropy" else 0), where):
                    yield name, kind, blob
        else:
            for k, p0 in enumerate([m.start() for m in re.finditer(b"\xff\xda", data)][1:4]):
                b = bytearray(data)
                b[p0 + int(rng.integers(-6, 14))] = int(rng.integers(0, 256))
                yield name, "sos%d" % k, bytes(b)


def one(item):
    name, kind, blob = item
    ref, rerr = reference(blob)
    if rerr in ("timeout", "crash", "no output"):
        return ("skip: reference " + rerr,)
    perr = product(blob)
    codes, is_float, oerr = O.decode_xt_status(blob)
    if oerr == 0 or (codes is None and oerr is None):
        # an alpha channel is read inside JPEG::Read behind the picture's codestreams: what is wrong with it fails the read
        # (... unless it is the alpha image's colour transformer that refuses: that waits for the first request for alpha pixels)
        aerr = O.alpha_read_error(blob)
        if aerr not in (None, 0) and (oerr == 0 or plain_oracle(blob)[1] == 0):
            codes, oerr = None, aerr
    if perr in DECLINED and rerr in (0, perr) or (perr in DECLINED and codes is None and oerr is None):
        return ("declined", name, kind, rerr, perr)  # (an XT file outside the accelerated subset: nothing to compare)
    if codes is None and oerr is None:  # no XT file (any more), or outside the XT restatement
        px, oerr = plain_oracle(blob)
        if oerr is None:
            return ("skip: outside the restatement", name, kind, rerr, perr)
    else:
        px = None if codes is None else (O.half_codes_to_float(codes) if is_float else codes)
    if oerr != rerr:
        return ("ORACLE code", name, kind, oerr, rerr, perr)
    if perr != oerr and perr not in DECLINED:
        return ("PRODUCT code", name, kind, perr, oerr)
    if rerr == 0 and (px.size != ref.size or not np.array_equal(px.reshape(ref.shape).astype(ref.dtype), ref)):
        return ("ORACLE pixels", name, kind, perr)
    return ("ok",) if perr not in DECLINED else ("declined", name, kind, rerr, perr)


def main():
    mode, seed = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
    work = list(work_list(mode, seed))
    with ThreadPoolExecutor(12) as ex:
        res = list(ex.map(one, work))
    print(mode, "seed", seed, dict(collections.Counter(r[0] for r in res)))
    for (name, kind, blob), r in zip(work, res):
        if r[0].isupper() or r[0].split()[0].isupper():
            print(" ", r)
            if keep:
                os.makedirs(keep, exist_ok=True)
                with open(os.path.join(keep, "%s_%d_%s_%s.jpg" % (mode, seed, name, kind)), "wb") as f:
                    f.write(blob)


if __name__ == "__main__":
    main()
