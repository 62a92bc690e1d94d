#!/usr/bin/env python3
"""GPU box (or any box with a device and oracle/_ref/jpeg): JPEG XT files written by the reference ENCODER with random switches, sizes
and inputs -- residual (8 / 12 bit, profile C), hidden bits in either frame, lossless (-ro, -Q 100, -rv), L-only (-R n alone), alpha
channels with and without their own residual -- decoded by the product on the device and compared, picture and alpha plane, with
the oracle (which the CPU suite pins against the reference decoder on the same kind of file).   N=400 SEED=1 python tools/xt_gpu_campaign.py"""
import collections
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libjpeg_amd import api, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

N = int(os.environ.get("N", "400"))
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))


def main():
    if not O.have_reference():
        sys.exit("needs oracle/_ref/jpeg (the encoder writes the inputs)")
    tmp = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    p = lambda n: os.path.join(tmp, n)  # noqa: E731
    dec = api.Decoder(0)
    count = collections.Counter()
    kernels = collections.Counter()
    bad = []
    made = 0
    while made < N:
        w, h = int(rng.integers(1, 140)), int(rng.integers(1, 100))
        kind = int(rng.integers(0, 5))
        img = synth.synth_image(w, h, int(rng.integers(1, 1 << 20)))
        hdr = synth.synth_hdr(w, h, int(rng.integers(1, 1 << 20))).astype("<f4")
        i16 = (np.clip(hdr / hdr.max(), 0, 1) ** 0.45 * 65535).astype(np.uint16)
        extra, ch = [], 3
        if kind == 0:
            O.write_ppm(p("in.ppm"), img); src = "in.ppm"
        elif kind == 1:
            O.write_ppm(p("in.pgm"), img[:, :, 0]); src, ch = "in.pgm", 1
        elif kind == 2:
            open(p("in.ppm"), "wb").write(b"P6\n%d %d\n65535\n" % (w, h) + i16.astype(">u2").tobytes()); src = "in.ppm"
        elif kind == 3:
            open(p("in.pgm"), "wb").write(b"P5\n%d %d\n65535\n" % (w, h) + i16[:, :, 0].astype(">u2").tobytes()); src, ch = "in.pgm", 1
        else:
            O.write_pfm(p("in.pfm"), hdr); src, extra = "in.pfm", ["-profile", "c"]
        mode = int(rng.integers(0, 6))
        args = ["-q", str(int(rng.integers(30, 97))), "-h"]
        if mode == 0:
            args += ["-r", "-Q", str(int(rng.integers(50, 98)))] + extra
            if rng.integers(0, 2): args += ["-r12"]
        elif mode == 1:
            args += ["-r", "-Q", str(int(rng.integers(50, 98))), "-ro"] + extra
        elif mode == 2:
            args += ["-r", "-Q", "100"] + extra
        elif mode == 3:
            if kind < 2:
                continue  # (-R alone needs more than eight bits)
            args += ["-R", str(int(rng.integers(1, 5)))]
        elif mode == 4:
            args += ["-r", "-Q", str(int(rng.integers(50, 98)))] + extra + ["-R", str(int(rng.integers(1, 4)))] if kind >= 2 else args + ["-r", "-Q", "90"] + extra
        else:
            args += (["-r", "-Q", str(int(rng.integers(50, 98)))] + extra) if rng.integers(0, 2) else []
        if "-r" in args and rng.integers(0, 3) == 0: args += ["-rR", str(int(rng.integers(1, 5)))]
        if "-r" in args and rng.integers(0, 3) == 0: args += ["-rv"]
        if rng.integers(0, 3) == 0: args += ["-v"]
        if ch == 3 and rng.integers(0, 2): args += ["-s", str(rng.choice(["1x1,2x2,2x2", "1x1,2x1,2x1", "1x1,1x2,1x2"]))]
        if rng.integers(0, 3) == 0: args += ["-z", str(int(rng.integers(1, 9)))]
        if "-r" in args and rng.integers(0, 4) == 0: args += ["-N"]
        if ch == 3 and kind == 0 and rng.integers(0, 5) == 0: args += ["-c"]
        alpha = mode == 5 or rng.integers(0, 4) == 0
        if alpha:
            a8 = synth.synth_image(w, h, int(rng.integers(1, 1 << 20)), channels=1).reshape(h, w)
            ak = int(rng.integers(0, 3))
            if ak == 0:
                O.write_ppm(p("a.pgm"), a8)
                aa = ["-al", p("a.pgm")]
                if rng.integers(0, 2):
                    aa += ["-aq", str(int(rng.integers(30, 95))), "-aQ", str(int(rng.integers(50, 98))), str(rng.choice(["-ar", "-ar12"]))]
                    if rng.integers(0, 3) == 0: aa += ["-arR", str(int(rng.integers(1, 4)))]
                    if rng.integers(0, 3) == 0: aa += ["-alo"]
            elif ak == 1:
                open(p("a.pgm"), "wb").write(b"P5\n%d %d\n65535\n" % (w, h) + (a8.astype(np.uint16) * 257).astype(">u2").tobytes())
                aa = ["-al", p("a.pgm"), "-aq", str(int(rng.integers(30, 95))), "-aQ", str(int(rng.integers(50, 98))), "-ar"]
            else:
                open(p("a.pfm"), "wb").write(b"Pf\n%d %d\n-1.0\n" % (w, h) + (a8.astype("<f4") / 255.0)[::-1].tobytes())
                aa = ["-al", p("a.pfm"), "-aq", str(int(rng.integers(30, 95))), "-aQ", str(int(rng.integers(50, 98))), "-ar"]
            args += ["-am", str(int(rng.integers(1, 4)))] + aa
        if os.path.exists(p("o.jpg")): os.remove(p("o.jpg"))
        r = subprocess.run([O.REF_BIN, *args, p(src), p("o.jpg")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=60)
        if r.returncode or b"failed" in r.stderr or b"Please" in r.stderr or not os.path.exists(p("o.jpg")):
            count["encode_failed"] += 1
            continue
        made += 1
        blob = open(p("o.jpg"), "rb").read()
        is_xt = b"SPEC" in blob or b"RESI" in blob
        try:
            if is_xt:
                want, _, oerr = O.decode_xt_status(blob)
            else:
                want, oerr, _ = O.decode_status(blob)
        except Exception as e:  # noqa: BLE001
            want, oerr = None, repr(e)
        try:
            info = dec.read(blob, entropy=os.environ.get("ENTROPY", "auto") if rng.integers(0, 2) else "host")  # (ENTROPY=prefer-gpu: the device entropy decoders however small the file)
            got = dec.reconstruct()
            perr = 0
        except api.MijpegError as e:
            got, perr = None, e.code
        what = (args, src, (w, h))
        if oerr is None and perr == -1034:
            count["declined"] += 1
            continue
        if oerr != 0 or perr != 0 or want is None:
            count["MISMATCH code"] += 1; bad.append((what, oerr, perr)); continue
        kernels[api.kernel_name(info)] += 1
        if not np.array_equal(np.asarray(got).reshape(-1).astype(np.uint16), np.asarray(want).reshape(-1).astype(np.uint16)):
            count["MISMATCH pixels"] += 1; bad.append((what, "pixels")); continue
        if b"ALFA" in blob:
            acodes, _, _, amode, _, aerr = O.decode_alpha(blob)
            a = dec.alpha_channel()
            if aerr != 0 or a is None:
                count["MISMATCH alpha code"] += 1; bad.append((what, aerr, a is None)); continue
            ag = a.reconstruct()
            if not np.array_equal(np.asarray(ag).reshape(-1).astype(np.uint16), acodes.reshape(-1)):
                count["MISMATCH alpha pixels"] += 1; bad.append((what, "alpha pixels")); continue
            count["ok with alpha"] += 1
        else:
            count["ok"] += 1
    print(f"{N} encoder-written files:", dict(count))
    print("reconstruction kernels:", dict(kernels))
    for b in bad[:30]:
        print("MISMATCH", b)


if __name__ == "__main__":
    main()
