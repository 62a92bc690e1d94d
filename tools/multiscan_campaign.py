#!/usr/bin/env python3
"""GPU box: random progressive pictures and JPEG XT files written by the reference encoder WITH restart markers (random sizes,
subsampling, qualities, restart intervals, scan scripts -v / -qv, hidden bits -R n / -rR n, progressive residuals -rv, 12-bit
input) -- the device decoder of progressive frames / hidden refinement scans (huffman_prog_kernel) against the host decoder,
coefficient plane by coefficient plane, and every eighth file's pixels against the oracle.
    N=300 SEED=1 [DAMAGE=1] [DUMP=dir] python tools/multiscan_campaign.py      (DAMAGE: every file corrupted, verdicts and coefficients
    compared; DUMP: streams whose verdicts differ are written there)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libjpeg_amd import api, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
import damage  # noqa: E402

N, SEED = int(os.environ.get("N", "200")), int(os.environ.get("SEED", "1"))
DAMAGE = bool(os.environ.get("DAMAGE"))
rng = np.random.default_rng(SEED)
dec, host = api.Decoder(0), api.Decoder(None)
SUBS = ["1x1,1x1,1x1", "1x1,2x2,2x2", "1x1,2x1,2x1", "1x1,1x2,1x2"]
stats = dict(device=0, declined=0, refused=0, coef_mismatch=0, pixel_mismatch=0, pixels_checked=0, verdict_mismatch=0)
for i in range(N):
    w, h = int(rng.integers(17, 700)), int(rng.integers(17, 500))
    z = int(rng.choice([1, 2, 3, 5, 8, 16, 40]))
    sub = SUBS[int(rng.integers(0, len(SUBS)))]
    kind = int(rng.integers(0, 4))
    try:
        if kind == 0:  # progressive picture
            args = [["-v"], ["-v", "-qv"], ["-v", "-h"]][int(rng.integers(0, 3))] + ["-q", str(int(rng.integers(30, 99))), "-s", sub, "-z", str(z)]
            data = O.reference_encode(synth.synth_image(w, h, SEED * 100000 + i), args)
        else:  # JPEG XT profile C with hidden bits and / or progressive codestreams
            args = ["-r", "-q", str(int(rng.integers(60, 95))), "-Q", str(int(rng.integers(60, 95))), "-h", "-profile", "c", "-r12", "-s", sub, "-z", str(z)]
            if rng.random() < 0.6:
                args += ["-rR", str(int(rng.integers(1, 5)))]
            if rng.random() < 0.4:
                args += ["-R", str(int(rng.integers(1, 5)))]
            if rng.random() < 0.3:
                args += ["-rv"]
            if rng.random() < 0.3:
                args += ["-v"]
            data = O.reference_encode_hdr(synth.synth_hdr(w, h, SEED * 100000 + i) * float(rng.choice([1.0, 4.0])), args)
    except Exception as e:  # noqa: BLE001 -- a switch combination the encoder itself turns down
        stats["refused"] += 1
        continue
    if DAMAGE:  # a seeded corruption of tests/damage.py: the answer must be the host walk's whatever the device decoder makes of the stream
        kind = damage.KINDS[int(rng.integers(0, len(damage.KINDS)))]
        data = damage.corrupt(data, kind, rng)
    try:
        fi = host.read(data, entropy="host")
        want = [host.coefficients(c) for c in range(fi.components)]
        want_r = [host.residual_coefficients(c) for c in range(host.xt_params().residual.components)] if fi.xt else []
        hv = 0
    except api.MijpegError as e:
        hv = e.code
    try:
        info = dec.read(data, entropy="prefer-gpu" if DAMAGE else "gpu")
        dv = 0
    except api.MijpegError as e:
        dv = e.code
    if DAMAGE and dv == 0 and dec.entropy_used != "gpu":
        stats["declined"] += 1  # (decoded by the host walk: compared below all the same)
    if dv == api.ERR_NOT_AVAILABLE:
        stats["declined"] += 1
        continue
    if dv != hv:
        stats["verdict_mismatch"] += 1
        print("verdict", i, args, hv, dv, flush=True)
        if os.environ.get("DUMP"):  # the stream, for a closer look
            os.makedirs(os.environ["DUMP"], exist_ok=True)
            with open(os.path.join(os.environ["DUMP"], "verdict_%d_%d.jpg" % (SEED, i)), "wb") as f:
                f.write(data)
        continue
    if dv:
        continue
    stats["device"] += 1
    ok = all(np.array_equal(dec.coefficients(c), want[c]) for c in range(info.components)) and \
        all(np.array_equal(dec.residual_coefficients(c), wr) for c, wr in enumerate(want_r))
    if not ok:
        stats["coef_mismatch"] += 1
        print("coefficients", i, w, h, args, flush=True)
        continue
    if i % 8 == 0 and not DAMAGE:
        stats["pixels_checked"] += 1
        got = dec.reconstruct()
        exp = O.decode_xt_status(data)[0] if info.xt else O.decode(data)
        if not np.array_equal(got, exp):
            stats["pixel_mismatch"] += 1
            print("pixels", i, w, h, args, flush=True)
print("summary", N, "files, seed", SEED, stats)
sys.exit(1 if stats["coef_mismatch"] or stats["pixel_mismatch"] or stats["verdict_mismatch"] else 0)
