#!/usr/bin/env python3
"""Regenerate tests/golden/: JPEG fixtures + the REAL reference's decode of each.

Run in the build container (needs oracle/_ref/jpeg, i.e. `make -C oracle ref` with /root/reference
present, and Pillow):   python tests/golden/make_golden.py

For every case it stores  <name>.jpg  and records in manifest.json the sha256 of the pixel bytes
the reference binary (cmd/reconstruct.cpp via `jpeg in.jpg out.ppm`) produced, the geometry, and
-- for the small cases -- the raw pixels in <name>.bin so a failing test can show a diff.
Large cases ("big") store no files: only sha256(jpeg bytes) + sha256(reference pixels); tests
regenerate the stream with the same deterministic recipe and use the entry if the bytes agree.
"""
import hashlib
import json
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from libjpeg_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def sha(b: bytes) -> str:
    return hashlib.sha256(b).hexdigest()


def insert_adobe(data: bytes, transform: int) -> bytes:
    """Insert an APP14 'Adobe' segment right after SOI (marker/adobemarker.cpp layout)."""
    seg = b"\xff\xee" + struct.pack(">H", 14) + b"Adobe" + struct.pack(">HHHB", 100, 0, 0, transform)
    return data[:2] + seg + data[2:]


def ref_sub(s):  # reference CLI takes SUBSAMPLING factors
    return ["-s", s]


CASES = []


def case(name, w, h, seed, enc, **kw):
    CASES.append(dict(name=name, w=w, h=h, seed=seed, enc=enc, **kw))


# BASELINE.json config 1: 512x512 4:4:4 Q75 baseline, reference-encoded
case("cfg1_512x512_444_q75_ref", 512, 512, 1234, "ref", args=["-bl", "-q", "75"])
# small-size analogues of configs 2/3 (reference encoder: Tq=0 everywhere; Pillow: Tq=1 chroma)
case("ref_80x48_420", 80, 48, 1, "ref", args=["-bl", "-q", "85", "-s", "1x1,2x2,2x2"])
case("ref_75x45_420_dri2", 75, 45, 2, "ref", args=["-bl", "-q", "85", "-s", "1x1,2x2,2x2", "-z", "2"])
case("ref_33x17_420_dri1", 33, 17, 3, "ref", args=["-bl", "-q", "60", "-s", "1x1,2x2,2x2", "-z", "1"])
case("ref_16x16_420", 16, 16, 4, "ref", args=["-bl", "-q", "90", "-s", "1x1,2x2,2x2"])
case("ref_100x9_422", 100, 9, 5, "ref", args=["-bl", "-q", "85", "-s", "1x1,2x1,2x1"])
case("ref_97x61_440", 97, 61, 6, "ref", args=["-bl", "-q", "85", "-s", "1x1,1x2,1x2"])
case("ref_97x61_411", 97, 61, 7, "ref", args=["-bl", "-q", "85", "-s", "1x1,4x1,4x1"])
case("ref_97x61_3x3", 97, 61, 8, "ref", args=["-bl", "-q", "85", "-s", "1x1,3x3,3x3", "-z", "3"])
case("ref_97x61_4x4", 97, 61, 9, "ref", args=["-bl", "-q", "85", "-s", "1x1,4x4,4x4"])
case("ref_64x64_2x4", 64, 64, 10, "ref", args=["-bl", "-q", "85", "-s", "1x1,2x4,2x4"])
case("ref_23x50_lumasub", 23, 50, 11, "ref", args=["-bl", "-q", "85", "-s", "2x2,1x1,1x1"])
case("ref_97x61_mixed", 97, 61, 12, "ref", args=["-bl", "-q", "85", "-s", "1x1,2x1,1x2"])
case("ref_120x88_sof1_420", 120, 88, 13, "ref", args=["-q", "85", "-s", "1x1,2x2,2x2", "-z", "4"])
case("ref_64x40_q2", 64, 40, 14, "ref", args=["-bl", "-q", "2", "-s", "1x1,2x2,2x2"])
case("ref_64x40_q100", 64, 40, 15, "ref", args=["-bl", "-q", "100"])
# number of lines in a DNL marker behind the first scan (SOF carries 0; codestream/entropyparser.cpp:204-249)
case("ref_97x61_420_dnl", 97, 61, 60, "ref", args=["-bl", "-n", "-q", "80", "-s", "1x1,2x2,2x2"])
case("ref_50x70_444_dnl_dri2", 50, 70, 61, "ref", args=["-bl", "-n", "-q", "80", "-z", "2"])
# merging specification boxes WITHOUT a residual codestream: the reference's encoder writes SPEC{OCON, LTRF = identity} (+ Adobe
# transform 0) for -c, and SPEC{OCON} for every grey scale picture that is not baseline; Tables::LTrafoTypeOf takes the box's
# transformation first (codestream/tables.cpp:1994-2021)
for _i, (_n, _s) in enumerate([("444", "1x1,1x1,1x1"), ("420", "1x1,2x2,2x2"), ("422", "1x1,2x1,2x1"), ("440", "1x1,1x2,1x2"),
                               ("411", "1x1,4x1,4x1"), ("3x3", "1x1,3x3,3x3"), ("lumasub", "2x2,1x1,1x1")]):
    case(f"refc_83x47_{_n}", 83, 47, 70 + _i, "ref", args=["-q", "85", "-c", "-s", _s] + (["-v"] if _i % 3 == 2 else []) + (["-z", "3"] if _i % 3 == 1 else []))
case("refc_64x40_444_prog", 64, 40, 78, "ref", args=["-q", "90", "-v", "-c"])
case("refspec_70x40_gray", 70, 40, 79, "refgray", args=["-q", "80"])
case("refspec_70x41_gray_prog_dri", 70, 41, 80, "refgray", args=["-q", "80", "-v", "-z", "4"])
case("pil_80x48_444", 80, 48, 20, "pil", quality=75, sub="444", dri=0)
case("pil_75x45_420_dri2", 75, 45, 21, "pil", quality=85, sub="420", dri=2)
case("pil_33x17_420_dri1", 33, 17, 22, "pil", quality=85, sub="420", dri=1)
case("pil_200x120_420_dri8", 200, 120, 23, "pil", quality=85, sub="420", dri=8)
case("pil_131x77_422", 131, 77, 24, "pil", quality=50, sub="422", dri=0)
case("pil_70x40_gray", 70, 40, 25, "pil", quality=80, sub="gray", dri=3)
case("pil_9x9_420", 9, 9, 26, "pil", quality=95, sub="420", dri=0)
case("pil_1x1_444", 1, 1, 27, "pil", quality=95, sub="444", dri=0)
case("pil_80x48_444_adobe0", 80, 48, 28, "pil", quality=75, sub="444", dri=0, adobe=0)
case("pil_80x48_420_adobe1", 80, 48, 29, "pil", quality=75, sub="420", dri=0, adobe=1)
# four components (Adobe CMYK): identity transformation, the reference writes one PGX raw file per component
case("pil_90x60_cmyk", 90, 60, 50, "cmyk", quality=85)
# progressive Huffman (SOF2): spectral selection + successive approximation
case("refprog_97x61_420", 97, 61, 40, "ref", args=["-v", "-q", "80", "-s", "1x1,2x2,2x2"])
case("refprog_64x64_444_dri5", 64, 64, 41, "ref", args=["-v", "-q", "80", "-z", "5"])
case("refprog_120x88_420_qv", 120, 88, 42, "ref", args=["-v", "-qv", "-q", "85", "-s", "1x1,2x2,2x2"])
case("refprog_75x45_420_dnl", 75, 45, 46, "ref", args=["-v", "-n", "-q", "80", "-s", "1x1,2x2,2x2"])
case("pilprog_200x130_422", 200, 130, 43, "pil", quality=90, sub="422", dri=0, progressive=True)
case("pilprog_75x45_420", 75, 45, 44, "pil", quality=75, sub="420", dri=0, progressive=True)
case("pilprog_70x40_gray", 70, 40, 45, "pil", quality=80, sub="gray", dri=0, progressive=True)
# JPEG XT profile C (BASELINE config 5, small analogues): HDR float input, reference encoder, reference PFM output
case("xt_64x48_444", 64, 48, 5, "xt", args=["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c", "-r12"])
case("xt_75x45_444", 75, 45, 6, "xt", args=["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c", "-r12"])
case("xt_129x71_420", 129, 71, 7, "xt", args=["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c", "-r12", "-s", "1x1,2x2,2x2"])
case("xt_200x120_420_q60", 200, 120, 8, "xt", args=["-r", "-q", "60", "-Q", "70", "-h", "-profile", "c", "-r12", "-s", "1x1,2x2,2x2"])
case("xt_33x17_422", 33, 17, 9, "xt", args=["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c", "-r12", "-s", "1x1,2x1,2x1"])
# ... with hidden refinement scans: -R n = n low bits of the legacy coefficients in FINE boxes (L table of 2^(8+n) entries),
# -rR n = n low bits of the residual coefficients in RFIN boxes (BASELINE config 5's "-rR 4" variant; 12 + 4 bits makes the
# reference's IDCT<4,QUAD> wrap in its second pass, which is part of what is pinned here)
case("xt_64x48_444_R4", 64, 48, 5, "xt", args=["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c", "-r12", "-R", "4"])
case("xt_75x45_444_rR4", 75, 45, 6, "xt", args=["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c", "-r12", "-rR", "4"])
case("xt_129x71_420_R2_rR3_dri3", 129, 71, 7, "xt", args=["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c", "-r12", "-R", "2", "-rR", "3",
                                                            "-s", "1x1,2x2,2x2", "-z", "3"])
case("xt_64x48_444_R1_rR1", 64, 48, 8, "xt", args=["-r", "-q", "70", "-Q", "80", "-h", "-profile", "c", "-r12", "-R", "1", "-rR", "1"])
case("xt_200x120_420_R3_rR4", 200, 120, 9, "xt", args=["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c", "-r12", "-R", "3", "-rR", "4",
                                                        "-s", "1x1,2x2,2x2"])
# hidden bits in the residual frame only, on a 4:2:0 legacy frame: BASELINE config 5's -rR variant in small, the shape
# fusedxtw420_kernel serves (round 3); one of them spans several 128-pixel tiles
case("xt_129x71_420_rR2", 129, 71, 41, "xt", args=["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c", "-r12", "-rR", "2", "-s", "1x1,2x2,2x2"])
case("xt_200x120_420_rR4", 200, 120, 42, "xt", args=["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c", "-r12", "-rR", "4", "-s", "1x1,2x2,2x2"])
case("xt_260x140_420_rR4_q60", 260, 140, 43, "xt", args=["-r", "-q", "60", "-Q", "75", "-h", "-profile", "c", "-r12", "-rR", "4", "-s", "1x1,2x2,2x2"])
case("xt_129x71_420_rR1_dri2", 129, 71, 44, "xt", args=["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c", "-r12", "-rR", "1", "-s", "1x1,2x2,2x2", "-z", "2"])
# plain 12-bit extended sequential (SOF1, P = 12): what the residual codestream of profile C is made of
case("p12_64x48_444", 64, 48, 30, "p12", args=["-q", "85"])
case("p12_120x90_420_dri3", 120, 90, 31, "p12", args=["-q", "85", "-s", "1x1,2x2,2x2", "-z", "3"])
# full-size configs 2 / 3 (hash-only)
case("big_4k_420_q85", 3840, 2160, 1234, "pil", quality=85, sub="420", dri=0, big=True)
case("big_4k_420_q85_dri8", 3840, 2160, 1234, "pil", quality=85, sub="420", dri=8, big=True)
case("big_8k_420_q85_dri8", 7680, 4320, 1234, "pil", quality=85, sub="420", dri=8, big=True)


UNSAMPLED = ["ref_80x48_420", "ref_97x61_3x3", "ref_23x50_lumasub", "ref_97x61_mixed", "pil_131x77_422", "pil_70x40_gray",
             "p12_120x90_420_dri3", "refprog_97x61_420"]


def synth_p12(w, h, seed):
    img = synth.synth_image(w, h, seed).astype(np.uint16) * 16
    return (img + (np.arange(w, dtype=np.uint16) % 16)[None, :, None]).astype(np.uint16)


def encode_p12(img, args):
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
        h, w = img.shape[:2]
        with open(d + "/in.ppm", "wb") as f:
            f.write(b"P6\n%d %d\n4095\n" % (w, h))
            f.write(img.astype(">u2").tobytes())
        subprocess.run([O.REF_BIN, *args, d + "/in.ppm", d + "/o.jpg"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        data = open(d + "/o.jpg", "rb").read()
        subprocess.run([O.REF_BIN, d + "/o.jpg", d + "/o.ppm"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        raw = open(d + "/o.ppm", "rb").read().split(b"\n", 3)[3]
        return data, np.frombuffer(raw, ">u2", w * h * 3).reshape(h, w, 3).astype(np.uint16)


def main():
    O.build()
    assert O.have_reference(), "oracle/_ref/jpeg missing: run `make -C oracle ref`"
    manifest = {}
    only = None
    if len(sys.argv) > 2 and sys.argv[1] == "--only":  # add cases to the committed set without touching the others
        only = set(sys.argv[2].split(","))
        with open(os.path.join(OUT, "manifest.json")) as f:
            manifest = json.load(f)
    for c in CASES:
        if only is not None and c["name"] not in only:
            continue
        if c["enc"] == "cmyk":
            import io
            import subprocess
            import tempfile

            from PIL import Image
            img4 = synth.synth_image(c["w"], c["h"], c["seed"], channels=4)
            buf = io.BytesIO()
            Image.fromarray(img4, "CMYK").save(buf, format="JPEG", quality=c["quality"])
            data = buf.getvalue()
            with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
                open(d + "/in.jpg", "wb").write(data)
                subprocess.run([O.REF_BIN, d + "/in.jpg", d + "/out"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                ref = np.stack([np.fromfile(d + "/out_%d.raw" % k, np.uint8).reshape(c["h"], c["w"]) for k in range(4)], -1)
            ent = dict(width=c["w"], height=c["h"], channels=4, seed=c["seed"], encoder="pil-cmyk", quality=c["quality"],
                       jpeg_sha256=sha(data), pixels_sha256=sha(ref.tobytes()), jpeg_bytes=len(data), pixels_file=c["name"] + ".bin")
            open(os.path.join(OUT, c["name"] + ".jpg"), "wb").write(data)
            open(os.path.join(OUT, c["name"] + ".bin"), "wb").write(ref.tobytes())
            manifest[c["name"]] = ent
            print(c["name"], len(data), ent["pixels_sha256"][:12])
            continue
        if c["enc"] in ("xt", "p12"):
            if c["enc"] == "xt":
                hdr = synth.synth_hdr(c["w"], c["h"], c["seed"]) * 4.0
                data = O.reference_encode_hdr(hdr, c["args"])
                ref = O.reference_decode_hdr(data)  # float32, what the reference CLI wrote as PFM
                assert ref.shape == (c["h"], c["w"], 3)
                pixel_bytes = np.ascontiguousarray(ref, "<f4").tobytes()
                kind = "xt_float32"
            else:
                data, ref = encode_p12(synth_p12(c["w"], c["h"], c["seed"]), c["args"])
                pixel_bytes = np.ascontiguousarray(ref, "<u2").tobytes()
                kind = "u16"
            ent = dict(width=c["w"], height=c["h"], channels=3, seed=c["seed"], encoder=c["enc"], kind=kind, args=c["args"],
                       jpeg_sha256=sha(data), pixels_sha256=sha(pixel_bytes), jpeg_bytes=len(data))
            with open(os.path.join(OUT, c["name"] + ".jpg"), "wb") as f:
                f.write(data)
            if len(pixel_bytes) <= 120000:
                with open(os.path.join(OUT, c["name"] + ".bin"), "wb") as f:
                    f.write(pixel_bytes)
                ent["pixels_file"] = c["name"] + ".bin"
            manifest[c["name"]] = ent
            print(c["name"], len(data), ent["pixels_sha256"][:12])
            continue
        img = synth.synth_image(c["w"], c["h"], c["seed"])
        if c["enc"] == "ref":
            data = O.reference_encode(img, c["args"])
        elif c["enc"] == "refgray":
            data = O.reference_encode(img[:, :, :1], c["args"])
        else:
            if c["sub"] == "gray":
                data = synth.encode_jpeg(img[..., 0], c["quality"], restart_mcus=c["dri"], progressive=c.get("progressive", False))
            else:
                data = synth.encode_jpeg(img, c["quality"], c["sub"], c["dri"], progressive=c.get("progressive", False))
            if "adobe" in c:
                data = insert_adobe(data, c["adobe"])
        ref = O.reference_decode(data)
        assert ref.shape[:2] == (c["h"], c["w"]), (c["name"], ref.shape)
        ent = dict(width=c["w"], height=c["h"], channels=int(ref.shape[2]), seed=c["seed"],
                   encoder=c["enc"], jpeg_sha256=sha(data), pixels_sha256=sha(ref.tobytes()),
                   jpeg_bytes=len(data))
        for k in ("args", "quality", "sub", "dri", "adobe"):
            if k in c:
                ent[k] = c[k]
        if c.get("big"):
            ent["big"] = True
        else:
            with open(os.path.join(OUT, c["name"] + ".jpg"), "wb") as f:
                f.write(data)
            if ref.size <= 120000:
                with open(os.path.join(OUT, c["name"] + ".bin"), "wb") as f:
                    f.write(ref.tobytes())
                ent["pixels_file"] = c["name"] + ".bin"
        manifest[c["name"]] = ent
        print(c["name"], len(data), ent["pixels_sha256"][:12])
    # reference `jpeg -U` (no upsampling: every component on its own grid, PGX raw files; cmd/reconstruct.cpp:208-306)
    import subprocess
    import tempfile
    for name in UNSAMPLED:
        with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
            with open(os.path.join(OUT, name + ".jpg"), "rb") as f:
                open(d + "/in.jpg", "wb").write(f.read())
            subprocess.run([O.REF_BIN, "-U", d + "/in.jpg", d + "/out"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            n = manifest[name]["channels"]
            manifest[name]["unsampled"] = [dict(header=open(d + "/out_%d.h" % k).read(), sha256=sha(open(d + "/out_%d.raw" % k, "rb").read()))
                                           for k in range(n)]
        print(name, "-U", [u["header"].strip() for u in manifest[name]["unsampled"]])
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
