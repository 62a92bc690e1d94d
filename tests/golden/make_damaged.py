#!/usr/bin/env python3
"""Regenerate tests/golden/damaged/: hand-made damaged codestreams + what the REAL reference does with each.

Run in the build container (needs oracle/_ref/jpeg, i.e. `make -C oracle ref` with /root/reference present):
    python tests/golden/make_damaged.py

Every case is a deterministic edit of one of the committed well-formed fixtures (tests/golden/<base>.jpg).  Stored:
<name>.jpg, and -- when the reference writes a picture -- <name>.bin with the pixel bytes it wrote
(cmd/reconstruct.cpp through `jpeg in.jpg out.ppm`); manifest.json records the verdict: "error" = 0 or the JPGERR_* code
the reference CLI printed (cmd/reconstruct.cpp:360-364), geometry and sha256 of the pixels.

What is exercised: the restart-marker resynchronisation of codestream/entropyparser.cpp:117-201 (marker behind / ahead of
the expected one, missing marker, zeroed MCUs of codestream/sequentialscan.cpp:416-420), the bit reader at the end of the
data (io/bitstream.cpp:56-137), the warn-and-continue of codestream/refinementscan.cpp:667 and header errors.
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import damage  # noqa: E402
from oracle import oracle as O  # noqa: E402

GOLDEN = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(GOLDEN, "damaged")


def base(name):
    with open(os.path.join(GOLDEN, name + ".jpg"), "rb") as f:
        return f.read()


def scan_starts(data):
    """Offsets of the first entropy-coded byte of every scan."""
    out = []
    p = 2
    while p + 4 <= len(data):
        if data[p] != 0xFF or data[p + 1] in (0x00, 0xFF):
            p += 1
            continue
        m = data[p + 1]
        if 0xD0 <= m <= 0xD9 or m == 0x01:
            p += 2
            continue
        ln = (data[p + 2] << 8) | data[p + 3]
        if m == 0xDA:
            out.append(p + 2 + ln)
        p += 2 + ln
    return out


def build_cases():
    cases = {}
    a = base("pil_200x120_420_dri8")  # 13 x 8 MCUs, an interval = 8 MCUs
    es = damage.entropy_start(a)
    ms = damage.restart_markers(a, es)
    d = bytearray(a); del d[ms[3]:ms[4]]
    cases["dri8_interval_dropped"] = ("pil_200x120_420_dri8", bytes(d), "marker 3 and the data behind it removed: the next marker is AHEAD")
    d = bytearray(a); d[ms[5]:ms[5]] = a[ms[4]:ms[5]]
    cases["dri8_interval_duplicated"] = ("pil_200x120_420_dri8", bytes(d), "interval 4 twice: the second copy's marker is BEHIND")
    d = bytearray(a); d[ms[6]:ms[8]] = a[ms[7]:ms[8]] + a[ms[6]:ms[7]]
    cases["dri8_intervals_swapped"] = ("pil_200x120_420_dri8", bytes(d), "intervals 6 and 7 exchanged")
    d = bytearray(a); del d[ms[2]:ms[2] + 2]
    cases["dri8_marker_removed"] = ("pil_200x120_420_dri8", bytes(d), "FF Dn of marker 2 removed, data kept")
    d = bytearray(a); d[ms[9] + 1] = 0xD0 + ((a[ms[9] + 1] - 0xD0 + 3) & 7)
    cases["dri8_marker_renumbered"] = ("pil_200x120_420_dri8", bytes(d), "marker 9 carries the number of marker 12")
    d = bytearray(a); del d[es + (len(a) - es) * 6 // 10:]
    cases["dri8_truncated"] = ("pil_200x120_420_dri8", bytes(d), "cut at 60 % of the entropy-coded data, no EOI")
    d = bytearray(a); q = ms[4] + 7; d[q:q + 3] = b"\xff\xff\xff"
    cases["dri8_ff_run"] = ("pil_200x120_420_dri8", bytes(d), "three FF bytes inside interval 4")
    b = base("ref_80x48_420")  # no restart markers
    es = damage.entropy_start(b)
    d = bytearray(b); d[es + (len(b) - es) // 2] ^= 0x5A
    cases["nodri_byte_flipped"] = ("ref_80x48_420", bytes(d), "one byte flipped in the middle of the only scan")
    d = bytearray(b); del d[es + (len(b) - es) // 3:]
    cases["nodri_truncated"] = ("ref_80x48_420", bytes(d), "cut at a third of the scan")
    c = base("refprog_64x64_444_dri5")  # progressive with restart intervals
    ss = scan_starts(c)
    d = bytearray(c); d[ss[-1] + 9] ^= 0xFF; d[ss[-1] + 10] ^= 0x3C
    cases["prog_last_scan_damaged"] = ("refprog_64x64_444_dri5", bytes(d), "two bytes of the last (refinement) scan damaged")
    ms = damage.restart_markers(c, ss[2])
    d = bytearray(c); del d[ms[1]:ms[2]]
    cases["prog_interval_dropped"] = ("refprog_64x64_444_dri5", bytes(d), "one restart interval of the third scan removed")
    e = base("ref_75x45_420_dri2")
    p = e.find(b"\xff\xc0")
    d = bytearray(e); d[p + 9] = 3  # Nf
    d[p + 11] = 0x55  # sampling factors 5x5 of the first component
    cases["header_bad_sampling"] = ("ref_75x45_420_dri2", bytes(d), "frame header with sampling factors 5x5")
    d = bytearray(e); del d[e.find(b"\xff\xc4"):e.find(b"\xff\xc4") + 2 + ((e[e.find(b"\xff\xc4") + 2] << 8) | e[e.find(b"\xff\xc4") + 3])]
    cases["header_missing_dht"] = ("ref_75x45_420_dri2", bytes(d), "first DHT segment removed")
    # a point transform in a SEQUENTIAL scan (Al = 2 in the SOS header; no encoder writes that, a flipped byte does): the
    # reference's sequential parser applies it (coefficients * 4) -- with and without restart markers
    for src_name in ("ref_80x48_420", "ref_75x45_420_dri2"):
        h = base(src_name)
        p = h.find(b"\xff\xda")
        d = bytearray(h)
        d[p + 2 + ((h[p + 2] << 8) | h[p + 3]) - 1] = 0x02  # Ah | Al, the last byte of the scan header
        cases["sequential_point_transform_" + ("dri2" if "dri2" in src_name else "nodri")] = (src_name, bytes(d), "Al = 2 in the header of a sequential scan")
    g = base("pil_70x40_gray")
    p = g.find(b"\xff\xc0")
    d = bytearray(g); d[p + 11] = 0xA3
    cases["header_gray_sampling_10x3"] = ("pil_70x40_gray", bytes(d), "single component with sampling factors 10x3: irrelevant, decodes")
    return cases


def main():
    if not O.have_reference():
        sys.exit("oracle/_ref/jpeg is missing: run `make -C oracle ref` first")
    os.makedirs(OUT, exist_ok=True)
    manifest = {}
    for name, (src, blob, what) in build_cases().items():
        px, err = O.reference_decode_status(blob)
        ent = {"base": src, "what": what, "error": err, "jpeg_sha256": hashlib.sha256(blob).hexdigest()}
        with open(os.path.join(OUT, name + ".jpg"), "wb") as f:
            f.write(blob)
        binfile = os.path.join(OUT, name + ".bin")
        if px is not None:
            ent.update(height=int(px.shape[0]), width=int(px.shape[1]), channels=int(px.shape[2]),
                       pixels_sha256=hashlib.sha256(px.tobytes()).hexdigest())
            with open(binfile, "wb") as f:
                f.write(px.tobytes())
        elif os.path.exists(binfile):
            os.remove(binfile)
        manifest[name] = ent
        print(f"{name:28s} reference: {'picture' if err == 0 else err}")
    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
