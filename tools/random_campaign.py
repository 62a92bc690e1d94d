#!/usr/bin/env python3
"""One-off differential campaign (longer than the test suite wants to be): N seeded random streams through the decoder object
(host and device entropy decoding, every kernel family) against the oracle, and random pictures through the encoder pipeline
against the oracle's forward restatement.  Prints a summary; exit code 1 on any difference."""
import collections
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from libjpeg_amd import api, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

N = int(os.environ.get("N", "1200"))
rng = np.random.default_rng(int(os.environ.get("SEED", "777")))
d = api.Decoder(0)
kernels, entropy, skipped, bad = collections.Counter(), collections.Counter(), 0, 0
t0 = time.time()
for t in range(N):
    big = rng.integers(0, 12) == 0
    w, h = (int(rng.integers(700, 2600)), int(rng.integers(500, 1600))) if big else (int(rng.integers(1, 700)), int(rng.integers(1, 500)))
    sub = ["444", "422", "420", "gray"][int(rng.integers(0, 4))]
    q = int(rng.choice([3, 20, 50, 75, 85, 95, 100]))
    dri = int(rng.choice([0, 0, 1, 3, 8, 40]))
    prog, opt = bool(rng.integers(0, 6) == 0), bool(rng.integers(0, 2))
    img = synth.synth_image(w, h, 5000 + t, channels=1 if sub == "gray" else 3)
    style = int(rng.integers(0, 4))
    if style == 0:
        img = rng.integers(0, 256, img.shape).astype(np.uint8)
    elif style == 1:  # saturated graphics
        img = (img > 128).astype(np.uint8) * 255
    try:
        data = synth.encode_jpeg(img, q, sub if sub != "gray" else "444", restart_mcus=dri, optimize=opt, progressive=prog)
    except OSError:
        skipped += 1
        continue
    exp = O.decode(data)
    for mode in ("host", "auto"):
        f = d.read(data, threads=int(rng.integers(1, 17)), entropy=mode)
        got = d.reconstruct()
        kernels[api.kernel_name(f)] += 1
        entropy[d.entropy_used] += 1
        if got.shape != exp.shape or not np.array_equal(got, exp):
            bad += 1
            print("DECODE DIFFERENCE", t, w, h, sub, q, dri, prog, opt, mode, flush=True)
    # every component 1 x 1 without a colour transformation (fused_flat_kernel): the 4:4:4 stream read as RGB stored as such,
    # and now and then a four-component (CMYK) file of the same picture
    if sub == "444" and not prog:
        info, planes = O.decode_coefficients(data)
        f = d.read(data, entropy="host")
        kernels[api.kernel_name(f, api.FLAG_NO_COLOR_TRANSFORM) + " (no transformation)"] += 1
        if not np.array_equal(d.reconstruct(api.FLAG_NO_COLOR_TRANSFORM), O.reconstruct(info, planes, use_ycbcr=0)):
            bad += 1
            print("DECODE DIFFERENCE (no transformation)", t, w, h, q, dri, flush=True)
        if t % 4 == 0:
            import io

            from PIL import Image
            buf = io.BytesIO()
            Image.fromarray(np.dstack([img, img[:, :, 1]]), "CMYK").save(buf, format="JPEG", quality=max(q, 5))
            cm = buf.getvalue()
            f = d.read(cm, entropy="host")
            kernels[api.kernel_name(f) + " (CMYK)"] += 1
            if not np.array_equal(d.reconstruct(), O.decode(cm)):
                bad += 1
                print("DECODE DIFFERENCE (CMYK)", t, w, h, q, flush=True)
    # encoder direction on the same picture (colour pictures only, the layouts the CLI offers)
    if sub != "gray" and t % 3 == 0:
        esub = ["444", "420", "422", "440", "411"][int(rng.integers(0, 5))]
        ri = int(rng.choice([0, 0, 2, 16]))
        enc = d.encode(img, q, esub, ri, opt)
        info, planes = O.decode_coefficients(enc)
        fwd = O.forward(info, img, 1)
        for c in range(3):
            nby, nbx = (info.ch[c] + 7) // 8, (info.cw[c] + 7) // 8
            if not np.array_equal(planes[c][:nby, :nbx], fwd[c][:nby, :nbx]):
                bad += 1
                print("ENCODE DIFFERENCE", t, w, h, esub, q, ri, opt, c, flush=True)
        kernels[api.kernel_name(d.read(enc, entropy="auto")) + " (own encoder's stream)"] += 1
        if not np.array_equal(d.reconstruct(), O.decode(enc)):
            bad += 1
            print("ENCODE->DECODE DIFFERENCE", t, w, h, esub, q, ri, flush=True)
        kernels["(encoded streams)"] += 1
# batches whose frames bring their own tables (mijpeg_batch.quant_dev): random shapes, layouts and qualities
NB = int(os.environ.get("N_BATCH", str(N // 20)))
if NB:
    import torch  # noqa: E402
batch_frames = 0
for t in range(NB):
    w, h = int(rng.integers(8, 900)), int(rng.integers(8, 600))
    sub = ["444", "422", "420", "gray", "440", "411"][int(rng.integers(0, 6))]
    nc = 1 if sub == "gray" else 3
    n = int(rng.integers(2, 9))
    quals = [int(rng.choice([5, 25, 50, 75, 85, 92, 98, 100])) for _ in range(n)]
    ri = int(rng.choice([1, 2, 5, 16]))
    imgs = [synth.synth_image(w, h, 9000 + 10 * t + i, channels=nc) for i in range(n)]
    if rng.integers(0, 3) == 0:
        imgs = [rng.integers(0, 256, im.shape).astype(np.uint8) for im in imgs]
    try:
        if sub in ("440", "411"):
            streams = [d.encode(im, q, sub, ri, bool(i & 1)) for i, (im, q) in enumerate(zip(imgs, quals))]
        else:
            streams = [synth.encode_jpeg(im, q, "444" if sub == "gray" else sub, restart_mcus=ri, optimize=bool(i & 1)) for i, (im, q) in enumerate(zip(imgs, quals))]
    except OSError:  # the test encoder refuses some noise pictures
        skipped += 1
        continue
    flags = int(rng.choice([0, 0, api.FLAG_FORCE_SAFE, api.FLAG_FORCE_GENERIC]))
    info = d.decode_batch_device(streams, min_intervals=1)
    row = w * nc
    out = torch.zeros((n, h, row), dtype=torch.uint8, device="cuda")
    d.reconstruct_batch_device(out.data_ptr(), h * row, row, flags)
    res = out.cpu().numpy().reshape(n, h, w, nc)
    kernels[api.kernel_name(info, flags) + " (batch, tables per frame)"] += 1
    for i in range(n):
        batch_frames += 1
        if not np.array_equal(res[i].squeeze(), O.decode(streams[i]).squeeze()):
            bad += 1
            print("BATCH DIFFERENCE", t, i, w, h, sub, quals, ri, flags, flush=True)
# damaged streams: random pictures and layouts, one seeded corruption each (tests/damage.py), through the decoder object with
# entropy = "prefer-gpu" (device where the stream still qualifies, the host walk with the reference's resynchronisation where
# not): return code and pixels against oracle/_ref/jpeg where it is present, else against the oracle's restatement
ND = int(os.environ.get("N_DAMAGED", str(N // 2)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import damage  # noqa: E402

use_ref = O.have_reference()
dstats = collections.Counter()
for t in range(ND):
    w, h = int(rng.integers(16, 900)), int(rng.integers(16, 600))
    sub = ["444", "422", "420", "gray"][int(rng.integers(0, 4))]
    q = int(rng.choice([20, 50, 75, 85, 95]))
    dri = int(rng.choice([0, 1, 2, 4, 8, 8, 20]))
    prog = bool(rng.integers(0, 8) == 0)
    img = synth.synth_image(w, h, 7000 + t, channels=1 if sub == "gray" else 3)
    try:
        data = synth.encode_jpeg(img, q, sub if sub != "gray" else "444", restart_mcus=dri, optimize=bool(rng.integers(0, 2)), progressive=prog)
    except OSError:
        skipped += 1
        continue
    kind = damage.KINDS[int(rng.integers(0, len(damage.KINDS)))]
    blob = damage.corrupt(data, kind, rng, "entropy" if rng.integers(0, 3) else "any")
    epx, eerr = damage.expected_of(blob, use_ref)
    if eerr is None:
        dstats["skip"] += 1
        continue
    try:
        d.read(blob, entropy="prefer-gpu")
        perr = 0
    except api.MijpegError as e:
        perr = e.code
    verdict = "ok"
    if perr != eerr:
        verdict = "code"
    elif perr == 0:
        got = d.reconstruct()
        if got.shape != epx.shape or not np.array_equal(got, epx):
            verdict = "pixels"
    dstats[verdict + (" (device entropy)" if perr == 0 and d.entropy_used == "gpu" else "")] += 1
    if verdict in ("code", "pixels"):
        bad += 1
        print("DAMAGED DIFFERENCE", t, w, h, sub, q, dri, prog, kind, verdict, eerr, perr, flush=True)
        os.makedirs(os.path.join(ROOT, "gpurun_out", "campaign_fail"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "campaign_fail", f"damaged_{t}_{kind}_{eerr}_{perr}.jpg"), "wb") as fo:
            fo.write(blob)
print(f"{ND} damaged streams against {'oracle/_ref/jpeg' if use_ref else 'the oracle'}:", dict(dstats))
# 12-bit frames (synth.to_12bit over random 8-bit streams, random delta scale: the fused 12-bit kernels and, beyond their
# gates, the unfused ones) and frames whose DC predictions leave 16 bits (damage.runaway_dc: int32 coefficient planes)
N12 = int(os.environ.get("N_12BIT", str(N // 4)))
k12 = collections.Counter()
for t in range(N12):
    w, h = int(rng.integers(1, 900)), int(rng.integers(1, 600))
    sub = ["444", "422", "420", "gray", "420", "gray"][int(rng.integers(0, 6))]
    q = int(rng.choice([30, 60, 85, 95]))
    dri = int(rng.choice([0, 0, 1, 8]))
    img = synth.synth_image(w, h, 9000 + t, channels=1 if sub == "gray" else 3)
    if rng.integers(0, 3) == 0:
        img = (img > 128).astype(np.uint8) * 255
    data = synth.to_12bit(synth.encode_jpeg(img, q, sub if sub != "gray" else "444", restart_mcus=dri), int(rng.choice([1, 5, 16, 16, 23, 64])))
    exp = O.decode16(data)
    f = d.read(data, entropy=["host", "prefer-gpu"][int(rng.integers(0, 2))])
    k12[api.kernel_name(f)] += 1
    got = d.reconstruct()
    if got.shape != exp.shape or not np.array_equal(got, exp) or not np.array_equal(d.reconstruct(api.FLAG_FORCE_GENERIC), exp):
        bad += 1
        print("12-BIT DIFFERENCE", t, w, h, sub, q, dri, list(f.range_max), flush=True)
print(f"{N12} 12-bit streams against the oracle:", dict(k12))
NR = int(os.environ.get("N_RUNAWAY", str(N // 10)))
kr = collections.Counter()
for t in range(NR):
    w, h = int(rng.integers(40, 700)), int(rng.integers(40, 500))
    sub = ["444", "422", "420"][int(rng.integers(0, 3))]
    base = synth.encode_jpeg(synth.synth_image(w, h, 9500 + t), 85, sub, restart_mcus=0)
    blob = damage.runaway_dc(base, 100 + t, drift=int(rng.choice([300, 900, 1800])))
    epx, eerr = damage.expected_of(blob, use_ref)
    try:
        f = d.read(blob, entropy="prefer-gpu")
        perr = 0
    except api.MijpegError as e:
        perr = e.code
    kr["wide" if perr == 0 and f.coef_wide else "16-bit"] += 1
    if perr != eerr or (perr == 0 and not np.array_equal(d.reconstruct(), np.asarray(epx).reshape(d.reconstruct().shape))):
        bad += 1
        print("RUNAWAY DIFFERENCE", t, w, h, sub, eerr, perr, flush=True)
print(f"{NR} runaway-DC streams against {'oracle/_ref/jpeg' if use_ref else 'the oracle'}:", dict(kr))
print(f"{N} random streams ({skipped} refused by the test encoder), {NB} batches with tables per frame ({batch_frames} frames), {time.time() - t0:.0f} s: {bad} differences")
print("reconstruction kernels:", dict(kernels))
print("entropy decoder used:", dict(entropy))
sys.exit(1 if bad else 0)
