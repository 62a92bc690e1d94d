"""JPEG::Read with JPGTAG_DECODER_STOP and PeekMarker / ReadMarker / SkipMarker (interface/jpeg.cpp:244-354, 505-645) of the
source-compatible class JPEG: tests/cxx/marker_calls.cpp -- the protocol of the reference's own marker injection test,
cmd/reconstruct.cpp:80-119 -- is compiled against libjpeg_amd's interface headers and run host-only (MIJPEG_DEVICE=-1).
Where the reference's objects are present (oracle/_ref/obj, build container) the SAME source is also linked against the
real reference library and the traces are compared line by line."""
import glob
import json
import os
import subprocess

import pytest

from conftest import GOLDEN_DIR, MANIFEST, ROOT

SRC = os.path.join(ROOT, "tests", "cxx", "marker_calls.cpp")
CASES = ["pil_200x120_420_dri8", "ref_75x45_420_dri2", "pil_70x40_gray", "pilprog_75x45_420", "refprog_64x64_444_dri5", "xt_64x48_444",
         "p12_64x48_444", "ref_97x61_3x3"]


def inject(data: bytes, extra: int, where: str = "all") -> bytes:
    """An APP9 segment behind the marker segments in front of the first SOS -- where = "pre": those in front of the frame
    header (what a client sees with STOP_IMAGE), "post": the frame header and those behind it (STOP_FRAME), "all" -- each
    followed by `extra` raw bytes (an EOI and half a DHT header among them) that only a client which removes them keeps
    away from the parser."""
    out = bytearray(data[:2])
    p, k = 2, 0
    seen_frame = False
    while data[p + 1] != 0xDA:
        m = data[p + 1]
        ln = (data[p + 2] << 8) | data[p + 3]
        out += data[p:p + 2 + ln]
        p += 2 + ln
        seen_frame = seen_frame or (0xC0 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC))
        nxt = data[p + 1]
        next_is_frame = 0xC0 <= nxt <= 0xCF and nxt not in (0xC4, 0xC8, 0xCC)
        if where == "all" or (where == "post" and seen_frame) or (where == "pre" and not seen_frame and not next_is_frame):
            out += b"\xff\xe9\x00\x06" + bytes([k, k + 1, 0xFF, 0xD9]) + (b"\xff\xd9\xff\xc4\x00" + bytes(range(16)))[:extra]
            k += 1
    return bytes(out + data[p:])


@pytest.fixture(scope="module")
def clients(tmp_path_factory):
    d = tmp_path_factory.mktemp("marker_calls")
    ours = str(d / "ours")
    subprocess.run(["g++", "-O1", "-w", "-I", os.path.join(ROOT, "libjpeg_amd", "csrc"), SRC, "-o", ours, "-L", os.path.join(ROOT, "libjpeg_amd"),
                    "-lmijpeg", "-Wl,-rpath," + os.path.join(ROOT, "libjpeg_amd")], check=True)
    ref = None
    objs = [o for o in glob.glob(os.path.join(ROOT, "oracle", "_ref", "obj", "**", "*.o"), recursive=True) if os.sep + "cmd" + os.sep not in o]
    refsrc = os.environ.get("LIBJPEG_REFERENCE", "/root/reference")
    if objs and os.path.isdir(refsrc):
        ref = str(d / "ref")
        subprocess.run(["g++", "-O1", "-w", "-DUSE_AUTOCONF", "-fno-exceptions", "-I", refsrc, "-I", os.path.join(ROOT, "oracle", "_ref", "gen"), SRC,
                        *objs, "-o", ref, "-lm"], check=True)
    return ours, ref


def run(exe, path, mode, extra=0):
    r = subprocess.run([exe, path, mode, str(extra)], capture_output=True, text=True, timeout=60, env=dict(os.environ, MIJPEG_DEVICE="-1"))
    return r.returncode, r.stdout.strip().splitlines()


def same_trace(a, b):
    """Line by line, no exceptions: since JPGFLAG_DECODER_STOP_SCAN is honoured (round 3) the stop loops of this library
    and of the reference stand at the same byte after every call."""
    return a == b


@pytest.mark.parametrize("mode", ["image", "frame"])
@pytest.mark.parametrize("name", CASES)
def test_stop_flags_and_marker_calls(clients, tmp_path, name, mode):
    ours, ref = clients
    ent = MANIFEST[name]
    src = os.path.join(GOLDEN_DIR, name + ".jpg")
    rc, lines = run(ours, src, mode)
    assert rc == 0 and f"info {ent['width']} {ent['height']} {ent['channels']}" in lines, lines
    assert lines[0].startswith("peek ff") and lines[-1] == "peek-after -1"
    with open(src, "rb") as f:
        data = f.read()
    # STOP_IMAGE stops at the segments in front of the frame header, STOP_FRAME at those behind it; the others are parsed by
    # the library itself, which skips a well-formed APP9 and trips over the garbage
    visible = "pre" if mode == "image" else "post"
    for extra, where in ((0, "all"), (9, visible), (9, "all")):
        inj = tmp_path / f"inj{extra}{where}.jpg"
        inj.write_bytes(inject(data, extra, where))
        rc, got = run(ours, str(inj), mode, extra)
        if extra == 0 or where == visible:
            assert rc == 0 and any(l.startswith("took") for l in got) and f"info {ent['width']} {ent['height']} {ent['channels']}" in got, got
        else:
            assert rc == 1 and any(l.startswith("error -10") for l in got), got
        if ref:
            rrc, exp = run(ref, str(inj), mode, extra)
            assert rrc == rc and same_trace(got, exp), (got, exp)
    if ref:
        rrc, exp = run(ref, src, mode)
        assert rrc == 0 and same_trace(lines, exp), (lines, exp)


def test_consumed_bytes_never_reach_the_parser(clients, tmp_path):
    """With STOP_IMAGE the client removes 'APP9 + 9 garbage bytes' (an EOI among them) behind every header segment: the
    decode must be that of the clean stream; without the client's help the same bytes end the parse."""
    ours, _ = clients
    name = "pil_200x120_420_dri8"
    with open(os.path.join(GOLDEN_DIR, name + ".jpg"), "rb") as f:
        data = f.read()
    inj = tmp_path / "inj.jpg"
    inj.write_bytes(inject(data, 9, "pre"))
    rc, lines = run(ours, str(inj), "image", 9)
    assert rc == 0 and "info 200 120 3" in lines
    rc, lines = run(ours, str(inj), "image", 0)  # the client takes the APP9 segments only: the garbage stays
    assert rc == 1, lines


SCAN_CASES = CASES + ["pilprog_200x130_422", "xt_200x120_420_R3_rR4", "xt_129x71_420_R2_rR3_dri3", "xt_64x48_444_R4", "pil_90x60_cmyk"]


@pytest.mark.parametrize("name", SCAN_CASES)
def test_stop_scan_returns_at_every_scan_header(clients, name):
    """JPEG::Read with JPGFLAG_DECODER_STOP_SCAN (interface/jpeg.cpp:310-353) in a loop until the data ends: one return per scan
    header with the stream positioned at that scan's entropy coded data (PeekMarker shows its first 16 bits), JPEG XT scans
    that live in boxes with the stream at the legacy frame's EOI; the image is complete after the last one.  The trace equals
    the real reference library's, call by call."""
    ours, ref = clients
    ent = MANIFEST[name]
    src = os.path.join(GOLDEN_DIR, name + ".jpg")
    rc, lines = run(ours, src, "scan")
    assert rc == 0 and f"info {ent['width']} {ent['height']} {ent['channels']}" in lines, lines
    peeks = [ln for ln in lines if ln.startswith("peek ")]
    assert peeks[-1] == "peek ffffffff" and len(peeks) >= 2
    if "prog" in name:
        assert len(peeks) >= 5  # one stop per progressive scan
    if ref:
        rrc, exp = run(ref, src, "scan")
        assert rrc == 0 and exp == lines, (lines, exp)


with open(os.path.join(GOLDEN_DIR, "stop_counts.json")) as _f:
    STOP_COUNTS = json.load(_f)


@pytest.mark.parametrize("mode", ["row", "mcu", "scanrow"])
@pytest.mark.parametrize("name", sorted(STOP_COUNTS))
def test_stop_row_and_stop_mcu_return_as_often_as_the_reference(clients, name, mode):
    """JPGFLAG_DECODER_STOP_ROW / _MCU (interface/jpeg.cpp:326-350): behind every scan header one return at the start of every MCU
    row, one behind every MCU of a row but its last -- walked over the grids of the scans this library has decoded by then
    (mijpeg_scan_grids).  What happens inside a scan cannot be told from the stream (the reference's input stands wherever its
    bit reader's buffer got to), so the trace is the NUMBER of returns: tests/golden/stop_counts.json holds the real library's
    (make_stop_counts.py), and where its objects are present the same client is run against it live."""
    ours, ref = clients
    rc, lines = run(ours, os.path.join(GOLDEN_DIR, name + ".jpg"), mode)
    assert rc == 0 and lines == STOP_COUNTS[name][mode], (lines, STOP_COUNTS[name][mode])
    if ref:
        rrc, exp = run(ref, os.path.join(GOLDEN_DIR, name + ".jpg"), mode)
        assert rrc == 0 and exp == lines
