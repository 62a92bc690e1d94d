"""JPEG XT files with an alpha channel (round 4 refused them whole).  The alpha channel is an image of its own in the
ALFA box -- one component, its own tables and scans, optionally a residual codestream (ARES) and hidden refinement scans (AFIN /
ARRF), its own merging specification ASPC with the compositing box AMUL -- that the reference reads inside JPEG::Read behind the
picture's codestreams (Image::ParseAlphaChannel, codestream/image.cpp:1337-1404, 1430-1460), describes in JPEG::GetInformation
(JPGTAG_ALPHA_MODE / _TAGLIST / _MATTE, interface/jpeg.cpp:919-951) and hands out beside the picture through
JPGTAG_BIH_ALPHAHOOK when JPGTAG_DECODER_INCLUDE_ALPHA is set (codestream/image.cpp:1087-1123); it does not composite.
tests/golden/xt_alpha/: 14 reference-written files with the reference decoder's picture and alpha plane.
CPU: oracle (oj_decode_alpha) against them and against the live binary; the product's host side (alpha decoder object, mode, matte,
error codes); -m gpu: picture and alpha plane through the C ABI, the command line, and the reference's own cmd/reconstruct.cpp client."""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT
from libjpeg_amd import api

DIR = os.path.join(GOLDEN_DIR, "xt_alpha")
with open(os.path.join(DIR, "manifest.json")) as _f:
    CASES = json.load(_f)
MODES = {"a8_opaque": 0, "a8_premultiplied": 2, "a8_matte": 3}
TMP = "/dev/shm" if os.path.isdir("/dev/shm") else None


def stream(name):
    with open(os.path.join(DIR, name + ".jpg"), "rb") as f:
        return f.read()


def picture(name):
    ent = CASES[name]
    return np.fromfile(os.path.join(DIR, name + ".pic.bin"), ent["picture_dtype"]).reshape(ent["height"], ent["width"], 3)


def alpha_plane(name):
    ent = CASES[name]
    if ent["alpha_dtype"] is None:
        return None
    return np.fromfile(os.path.join(DIR, name + ".alpha.bin"), ent["alpha_dtype"]).reshape(ent["height"], ent["width"])


def oracle_picture(oracle, data):
    codes, is_float, err = oracle.decode_xt_status(data)
    assert err == 0
    return codes


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_against_the_reference_decoder(oracle, name):
    codes, is_float, out_max, mode, matte, err = oracle.decode_alpha(stream(name))
    assert err == 0 and mode == MODES.get(name, 1) and matte == ((10, 200, 30) if name == "a8_matte" else (0, 0, 0))
    want = alpha_plane(name)
    if want is not None:  # (method 0: the reference's command line writes no alpha file)
        assert is_float == CASES[name]["alpha_float"] and out_max == (255 if want.dtype == np.uint8 else 65535)
        assert np.array_equal(codes, want.astype(np.uint16)), name
    assert np.array_equal(oracle_picture(oracle, stream(name)), picture(name).astype(np.uint16)), name


@pytest.mark.parametrize("name", sorted(CASES))
def test_host_side(oracle, name):
    """The alpha channel's decoder object: one component, the size of the picture, mode and matte of the AMUL box."""
    d = api.Decoder(None)
    f = d.read(stream(name))
    a = d.alpha_channel()
    assert a is not None and a.info.components == 1 and (a.info.width, a.info.height) == (f.width, f.height)
    assert d.alpha_info() == (MODES.get(name, 1), (10, 200, 30) if name == "a8_matte" else (0, 0, 0))
    want = alpha_plane(name)
    if want is not None:
        assert a.info.sample_bytes == want.dtype.itemsize and bool(a.info.is_float) == CASES[name]["alpha_float"]
    if "residual" in name or name in ("a8_beside_int_xt", "af_beside_hdr", "a8_openloop"):
        assert a.info.xt == 1 and a.xt_params().residual.components == 1 and a.xt_params().no_residual == 0
    d.close()


def test_a_file_without_alpha_has_no_alpha_channel():
    d = api.Decoder(None)
    with open(os.path.join(GOLDEN_DIR, "xt_int8", "enc_444.jpg"), "rb") as f:
        d.read(f.read())
    assert d.alpha_channel() is None
    with pytest.raises(api.MijpegError) as e:
        d.alpha_info()
    assert e.value.code == -1031
    d.close()


def test_asking_for_alpha_leaves_no_error_behind_and_children_die_with_their_owner():
    """mijpeg_has_alpha is a query (JPEG::GetInformation asks on every file): the object's last error stays what it was.  The
    alpha decoder a parent handed out is the parent's: closing the parent invalidates the binding's child object."""
    import ctypes as C

    L = api.lib()
    d = api.Decoder(None)
    with open(os.path.join(GOLDEN_DIR, "xt_int8", "enc_444.jpg"), "rb") as f:
        d.read(f.read())
    assert L.mijpeg_has_alpha(d._h) == 0
    msg = C.c_char_p()
    assert L.mijpeg_last_error(d._h, C.byref(msg)) == 0 and msg.value is None
    d.close()
    d = api.Decoder(None)
    d.read(stream("a8_matte"))
    assert L.mijpeg_has_alpha(d._h) == 1
    a = d.alpha_channel()
    assert a is not None and a._h
    d.close()
    assert not a._h  # no dangling handle: a call on it fails in the binding, not in freed memory
    with pytest.raises(api.MijpegError):
        a._check(-1024)


def _segments(data, box_type):
    out, i = [], 2
    while i + 4 <= len(data) and data[i] == 0xFF and data[i + 1] != 0xDA:
        ln = (data[i + 2] << 8) | data[i + 3]
        if data[i + 1] == 0xEB and data[i + 4:i + 6] == b"JP" and data[i + 16:i + 20] == box_type:
            out.append((i, ln))
        i += 2 + ln
    return out


def damaged_alpha(name):
    """kind -> stream: what is wrong with the alpha codestream fails the read (it is parsed inside JPEG::Read), an incomplete ALFA
    box or a legacy codestream without EOI leaves a file without alpha channel"""
    data = stream(name)
    (off, ln), = _segments(data, b"ALFA")
    out = {}
    b = bytearray(data)
    b[off + 20] = 0x00  # SOI of the alpha codestream gone: "Alpha channel codestream is invalid, SOI marker missing."
    out["no_soi"] = bytes(b)
    seg = data[off:off + 2 + ln]
    sof = seg.find(b"\xff\xc0") if b"\xff\xc0" in seg else seg.find(b"\xff\xc1")
    b = bytearray(data)
    b[off + sof + 8] ^= 0x01  # the alpha frame's width: "dimensions do not match"
    out["width"] = bytes(b)
    sos = seg.find(b"\xff\xda")
    b = bytearray(data)
    b[off + sos + 20:off + sos + 22] = b"\xff\xc4"  # a DHT marker in the alpha scan
    out["scan"] = bytes(b)
    b = bytearray(data)
    lbox = int.from_bytes(b[off + 12:off + 16], "big") + 1  # the box never completes: no alpha channel
    b[off + 12:off + 16] = lbox.to_bytes(4, "big")
    out["lbox"] = bytes(b)
    out["no_eoi"] = data[:-2]  # the reference never gets to the alpha channel
    # the alpha merging specification gone (its segment's first byte swallowed by the one in front of it, tools/box_campaign.py): no
    # compositing method -> JPEG::GetInformation reports no alpha channel and nobody asks for one; with an ARES box left behind the
    # alpha image's transformer would refuse (R transformation "zero" beside a residual, -1024) -- at a request that never comes
    (aoff, aln), = _segments(data, b"ASPC")
    out["no_aspc"] = data[:aoff] + data[aoff + 2 + aln:]
    return out


@pytest.mark.parametrize("name", ["a8", "a8_residual"])
def test_damaged_alpha_channels(oracle, name):
    for kind, blob in damaged_alpha(name).items():
        d = api.Decoder(None)
        try:
            d.read(blob)
            # (what the command line goes by: an alpha channel whose specification names a compositing method other than "opaque")
            perr, has = 0, d.alpha_channel() is not None and d.alpha_info()[0] >= 1
        except api.MijpegError as e:
            perr, has = e.code, False
        d.close()
        if oracle.have_reference():
            with tempfile.TemporaryDirectory(dir=TMP) as t:
                src, dst, adst = os.path.join(t, "i.jpg"), os.path.join(t, "o.ppm"), os.path.join(t, "a.pgm")
                open(src, "wb").write(blob)
                r = subprocess.run([oracle.REF_BIN, "-al", adst, src, dst], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
                import re
                m = re.search(rb"failed - error (-?\d+)", r.stderr)
                rerr, rhas = (int(m.group(1)) if m else 0), os.path.exists(adst) and os.path.getsize(adst) > 20
            assert perr == rerr, (name, kind, perr, rerr, r.stderr[-200:])
            assert has == rhas, (name, kind)
        if kind in ("lbox", "no_eoi", "no_aspc"):
            assert perr == 0 and not has, (name, kind)
            if kind == "no_aspc":
                assert oracle.alpha_read_error(blob) == 0
        else:
            assert perr == -1038, (name, kind, perr)


def _live_cases(oracle, count, seed):
    from libjpeg_amd import synth

    rng = np.random.default_rng(seed)
    made = 0
    while made < count:
        w, h = int(rng.integers(1, 70)), int(rng.integers(1, 60))
        with tempfile.TemporaryDirectory(dir=TMP) as d:
            p = lambda n: os.path.join(d, n)  # noqa: E731
            img = synth.synth_image(w, h, int(rng.integers(1, 1 << 20)))
            a8 = synth.synth_image(w, h, int(rng.integers(1, 1 << 20)), channels=1).reshape(h, w)
            oracle.write_ppm(p("in.ppm"), img)
            kind = int(rng.integers(0, 3))
            if kind == 0:
                oracle.write_ppm(p("a.pgm"), a8)
                aargs = ["-al", p("a.pgm")]
                if rng.integers(0, 2):
                    aargs += ["-aq", str(int(rng.integers(30, 95))), "-aQ", str(int(rng.integers(50, 98))), str(rng.choice(["-ar", "-ar12"]))]
                    if rng.integers(0, 3) == 0:
                        aargs += ["-arR", str(int(rng.integers(1, 4)))]
                    if rng.integers(0, 3) == 0:  # lossless / near-lossless alpha: the residual scan type in the ARES box (-h: its tables)
                        aargs += [str(rng.choice(["-alo", "-aQ"])), "-h"]
                        if aargs[-2] == "-aQ":
                            aargs.insert(-1, "100")
                elif rng.integers(0, 3) == 0:
                    aargs += ["-aR", str(int(rng.integers(1, 4)))]
            elif kind == 1:
                with open(p("a.pgm"), "wb") as f:
                    f.write(b"P5\n%d %d\n65535\n" % (w, h) + (a8.astype(np.uint16) * 257).astype(">u2").tobytes())
                aargs = ["-al", p("a.pgm"), "-aq", str(int(rng.integers(30, 95))), "-aQ", str(int(rng.integers(50, 98))), "-ar", "-h"]
            else:
                with open(p("a.pfm"), "wb") as f:
                    f.write(b"Pf\n%d %d\n-1.0\n" % (w, h) + (a8.astype("<f4") / 255.0)[::-1].tobytes())
                aargs = ["-al", p("a.pfm"), "-aq", str(int(rng.integers(30, 95))), "-aQ", str(int(rng.integers(50, 98))), "-ar", "-h"]
            args = ["-q", str(int(rng.integers(30, 95))), "-am", str(int(rng.integers(1, 4)))] + aargs
            if rng.integers(0, 2):
                args += ["-s", str(rng.choice(["1x1,2x2,2x2", "1x1,2x1,2x1"]))]
            if rng.integers(0, 3) == 0:
                args += ["-r", "-Q", str(int(rng.integers(60, 98)))]
            r = subprocess.run([oracle.REF_BIN, *args, p("in.ppm"), p("o.jpg")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            if r.returncode or b"failed" in r.stderr or b"Please" in r.stderr:
                continue
            blob = open(p("o.jpg"), "rb").read()
            if b"ALFA" not in blob:
                continue
            r = subprocess.run([oracle.REF_BIN, "-al", p("ao"), p("o.jpg"), p("po")], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            assert r.returncode == 0 and b"failed" not in r.stderr, (args, r.stderr)
            magic = open(p("ao"), "rb").read(2)
            alpha = oracle.read_pfm_reference(p("ao")).astype("<f2").view("<u2").reshape(h, w) if magic == b"Pf" else oracle.read_pnm_any(p("ao")).reshape(h, w)
            pic = oracle.read_pnm_any(p("po")).reshape(h, w, 3)
        made += 1
        yield blob, pic, alpha, (w, h, args)


def test_oracle_against_live_reference(oracle):
    if not oracle.have_reference():
        pytest.skip("oracle/_ref/jpeg not built")
    n = 0
    for blob, pic, alpha, what in _live_cases(oracle, 60, 20260927):
        codes, is_float, out_max, mode, matte, err = oracle.decode_alpha(blob)
        assert err == 0, what
        assert np.array_equal(codes, alpha.astype(np.uint16)), what
        assert np.array_equal(oracle_picture(oracle, blob), pic.astype(np.uint16)), what
        n += 1
    assert n == 60


# ------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def dec():
    d = api.Decoder(0)
    yield d
    d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("entropy", ["host", "auto"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_picture_and_alpha_equal_the_reference(oracle, dec, name, entropy):
    dec.read(stream(name), entropy=entropy)
    out = dec.reconstruct()
    assert np.array_equal(out.reshape(picture(name).shape), picture(name)), name
    a = dec.alpha_channel()
    assert a is not None
    plane = a.reconstruct()
    want = alpha_plane(name)
    if want is None:  # method 0: no file from the reference's command line; the plane itself is the oracle's
        want, _, _, _, _, err = oracle.decode_alpha(stream(name))
        assert err == 0
    assert np.array_equal(plane.reshape(want.shape), want.astype(plane.dtype)), name


@pytest.mark.gpu
@pytest.mark.parametrize("client", ["cli", "dropin"])
@pytest.mark.parametrize("name", ["a8", "a8_420", "a8_matte", "a8_residual12", "a16_residual", "af_beside_hdr", "a8_opaque"])
def test_gpu_clients_write_the_references_files(oracle, tmp_path, name, client):
    """`jpeg -al alpha.pgm in.jpg out.ppm`: this repository's command line, and the reference's own cmd/reconstruct.cpp client
    compiled unchanged against libmijpeg.so (oracle/_ref/jpeg_dropin: AlphaHook of cmd/bitmaphook.cpp, JPGTAG_ALPHA_TAGLIST)."""
    exe = os.path.join(ROOT, "libjpeg_amd", "bin", "jpeg") if client == "cli" else os.path.join(ROOT, "oracle", "_ref", "jpeg_dropin")
    if not os.path.exists(exe):
        pytest.skip(exe + " not built")
    src, dst, adst = tmp_path / "in.jpg", tmp_path / "out.pic", tmp_path / "out.alpha"
    src.write_bytes(stream(name))
    subprocess.run([exe, "-al", str(adst), str(src), str(dst)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

    def samples(path):
        with open(path, "rb") as f:
            magic = f.read(2)
        return oracle.read_pfm_reference(str(path)).astype("<f2").view("<u2") if magic in (b"PF", b"Pf") else oracle.read_pnm_any(str(path))

    assert np.array_equal(samples(dst).reshape(picture(name).shape), picture(name)), name
    want = alpha_plane(name)
    if want is None:
        assert not adst.exists() or adst.stat().st_size == 0
    else:
        assert np.array_equal(samples(adst).reshape(want.shape), want), name


def test_the_alpha_frame_header_is_judged_before_its_scans(oracle):
    """Image::ParseAlphaChannel compares the alpha image's dimensions right behind its frame header (codestream/image.cpp:1366-1380): a
    damaged width is -1038 at the read, whatever the scan's data would do to a frame of that size; and a byte that turns the SOF
    marker into one of a coding process outside this library in front of a header that no longer fits is the reference's "frame
    header marker size is invalid" (-1038), not a refusal that waits for the request (tools/box_campaign.py r5)."""
    data = stream("a16_residual")
    (off, ln), = _segments(data, b"ALFA")
    seg = data[off:off + 2 + ln]
    sof = seg.find(b"\xff\xc1")
    wide = bytearray(data)
    wide[off + sof + 7] = 0x1A  # the high byte of the alpha frame's width
    tall = bytearray(data)
    tall[off + sof + 5] = 0x0A  # ... of its height (a frame its scan's data could never fill: -1025 if the scan were looked at first)
    other = bytearray(stream("a8_420"))
    (off2, ln2), = _segments(bytes(other), b"ALFA")
    sof2 = bytes(other[off2:off2 + 2 + ln2]).find(b"\xff\xc1")
    assert sof2 > 0
    shifted = bytes(other[:off2 + sof2 + 1]) + b"\xb3" + bytes(other[off2 + sof2 + 1:off2 + 2 + ln2 - 1]) + bytes(other[off2 + 2 + ln2:])  # FF B3 C0 ..: the box keeps its size
    for blob in (bytes(wide), bytes(tall), shifted):
        assert oracle.alpha_read_error(blob) == -1038
        if oracle.have_reference():
            assert oracle.reference_decode_status(blob)[1] == -1038
        d = api.Decoder(None)
        with pytest.raises(api.MijpegError) as e:
            d.read(blob)
        assert e.value.code == -1038
        d.close()


def _alpha_with_reference(oracle, blob):
    """-> (the reference's code, its alpha plane or None) through `jpeg -al`"""
    import re
    with tempfile.TemporaryDirectory(dir=TMP) as t:
        src, dst, adst = os.path.join(t, "i.jpg"), os.path.join(t, "o.ppm"), os.path.join(t, "a.pgm")
        open(src, "wb").write(blob)
        r = subprocess.run([oracle.REF_BIN, "-al", adst, src, dst], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        m = re.search(rb"failed - error (-?\d+)", r.stderr)
        plane = oracle.read_pnm_any(adst) if (not m and os.path.exists(adst) and os.path.getsize(adst) > 20) else None
        return (int(m.group(1)) if m else 0), plane


def alpha_codestream_without_eoi(name="a8_residual_hidden"):
    data = bytearray(stream(name))
    (off, ln), = _segments(bytes(data), b"ALFA")
    assert data[off + ln:off + ln + 2] == b"\xff\xd9"
    data[off + ln + 1] = 0x44
    return bytes(data)


def test_the_alpha_residual_is_read_however_the_alpha_codestream_ends(oracle):
    """The main image's residual codestream is reached at the legacy EOI only; the ALPHA image's is turned to by the outer image's
    trailer when Image::ParseAlphaChannel has no more scans to give -- EOI, end of the box, a marker, anything
    (codestream/image.cpp:1440-1462).  An alpha codestream whose EOI is gone still merges its residual (tools/box_campaign.py r5:
    2044 of 2183 alpha samples differed); and a residual codestream that does not parse fails the read whatever the alpha merging
    specification says -- one this library declines included."""
    no_eoi = alpha_codestream_without_eoi()
    codes, is_float, out_max, mode, matte, err = oracle.decode_alpha(no_eoi)
    assert err == 0
    if oracle.have_reference():
        rerr, plane = _alpha_with_reference(oracle, no_eoi)
        assert rerr == 0 and np.array_equal(plane.reshape(codes.shape), codes)
    d = api.Decoder(None)
    d.read(no_eoi)
    a = d.alpha_channel()
    assert a is not None and d.alpha_info()[0] == 1
    d.close()
    # a specification outside the accelerated path (the integer DCT in the residual domain) beside an ARES codestream whose DHT is damaged
    data = bytearray(stream("a8_openloop"))
    (soff, sln), = _segments(bytes(data), b"ASPC")
    i = bytes(data).index(b"RDCT", soff)
    assert data[i + 4] == 0x00
    data[i + 4] = 0x20
    declined = bytes(data)
    if oracle.have_reference():
        assert oracle.reference_decode_status(declined)[1] == 0
    d = api.Decoder(None)
    d.read(declined)
    assert d.alpha_channel() is None  # (declined: the picture reads, the alpha request is refused)
    d.close()
    (roff, rln), = _segments(declined, b"ARES")
    j = declined.index(b"\xff\xc4", roff)
    broken = declined[:j + 4] + b"\x7f" + declined[j + 5:]  # table class / index byte of the ARES codestream's first DHT
    if oracle.have_reference():
        assert oracle.reference_decode_status(broken)[1] == -1038
    assert oracle.alpha_read_error(broken) == -1038
    d = api.Decoder(None)
    with pytest.raises(api.MijpegError) as e:
        d.read(broken)
    assert e.value.code == -1038
    d.close()


def marker_in_the_alpha_scan(name="a8_matte"):
    """three bytes of the alpha codestream's entropy coded data become FF FF FF in front of a byte that makes a marker of them"""
    data = bytearray(stream(name))
    (off, ln), = _segments(bytes(data), b"ALFA")
    seg = bytes(data[off:off + 2 + ln])
    sos = seg.find(b"\xff\xda")
    data[off + sos + 40:off + sos + 43] = b"\xff\xff\xff"
    data[off + sos + 43] = 0xB1
    return bytes(data)


def test_a_marker_behind_the_alpha_scan_is_no_second_frame(oracle):
    """Image::ParseAlphaChannel hands the SAME frame back when the alpha image's trailer finds a marker (codestream/image.cpp:1386-1397,
    like ParseResidualStream for the residual codestream): FF B1 in the alpha scan's data ends the scan, is no "double frame header", and
    the picture and the alpha plane -- the rest of the scan zero bits -- come out (tools/box_campaign.py r5: oracle and product said -1038)."""
    blob = marker_in_the_alpha_scan()
    codes, is_float, out_max, mode, matte, err = oracle.decode_alpha(blob)
    assert err == 0 and codes is not None and mode == 3
    if oracle.have_reference():
        with tempfile.TemporaryDirectory(dir=TMP) as t:
            src, dst, adst = os.path.join(t, "i.jpg"), os.path.join(t, "o.ppm"), os.path.join(t, "a.pgm")
            open(src, "wb").write(blob)
            r = subprocess.run([oracle.REF_BIN, "-al", adst, src, dst], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            assert r.returncode == 0 and b"failed" not in r.stderr
            assert np.array_equal(oracle.read_pnm_any(adst).reshape(codes.shape), codes)
    d = api.Decoder(None)
    d.read(blob)
    assert d.alpha_channel() is not None and d.alpha_info()[0] == 3
    d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("make", [marker_in_the_alpha_scan, alpha_codestream_without_eoi])
def test_gpu_alpha_plane_of_damaged_alpha_codestreams(oracle, dec, make):
    blob = make()
    dec.read(blob)
    a = dec.alpha_channel()
    assert np.array_equal(a.reconstruct().reshape(a.info.height, a.info.width), oracle.decode_alpha(blob)[0])


@pytest.mark.gpu
def test_gpu_alpha_stripes_through_display_rect(oracle, dec):
    """The alpha image has row cursors of its own: eight-line stripes through mijpeg_display_rect like the command line asks."""
    name = "a8_residual"
    dec.read(stream(name))
    a = dec.alpha_channel()
    h, w = a.info.height, a.info.width
    canvas = np.zeros((1, h, w), np.uint8)  # (planar: one plane per component)
    for y in range(0, h, 8):
        a.display_rect(canvas, 0, y, w - 1, min(h, y + 8) - 1, bm_height=8 + y)
    assert np.array_equal(canvas.reshape(h, w), alpha_plane(name))


@pytest.mark.gpu
def test_gpu_live_sweep_against_the_oracle(oracle, dec):
    if not oracle.have_reference():
        pytest.skip("needs the reference encoder (build container)")
    for blob, pic, alpha, what in _live_cases(oracle, 30, 11):
        dec.read(blob)
        assert np.array_equal(dec.reconstruct().reshape(pic.shape), pic), what
        plane = dec.alpha_channel().reconstruct()
        assert np.array_equal(plane.reshape(alpha.shape), alpha.astype(plane.dtype)), what


@pytest.mark.gpu
def test_gpu_alpha_at_4k(oracle, dec):
    """3840 x 2160 4:2:0 picture with an 8-bit alpha channel: both against the oracle."""
    if not oracle.have_reference():
        pytest.skip("needs the reference encoder (build container)")
    from libjpeg_amd import synth

    w, h = 3840, 2160
    with tempfile.TemporaryDirectory(dir=TMP) as d:
        oracle.write_ppm(os.path.join(d, "in.ppm"), synth.synth_image(w, h, 1234))
        oracle.write_ppm(os.path.join(d, "a.pgm"), synth.synth_image(w, h, 77, channels=1).reshape(h, w))
        subprocess.run([oracle.REF_BIN, "-q", "85", "-s", "1x1,2x2,2x2", "-z", "8", "-al", os.path.join(d, "a.pgm"), os.path.join(d, "in.ppm"),
                        os.path.join(d, "o.jpg")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        data = open(os.path.join(d, "o.jpg"), "rb").read()
    codes, _, _, _, _, err = oracle.decode_alpha(data)
    assert err == 0
    dec.read(data)
    assert np.array_equal(dec.reconstruct(), oracle.decode(data))
    plane = dec.alpha_channel().reconstruct()
    assert np.array_equal(plane.reshape(h, w), codes.astype(plane.dtype))


def _alpha_residual_with_a_runaway_coefficient():
    """a8_residual_hidden with one byte of the alpha residual codestream's entropy coded data changed: its decode leaves 52 241 in
    one block of a 9-bit frame (8 bits + one hidden bit) -- found by tools/xt_gpu_damage_campaign.py, seed 2002."""
    data = bytearray(stream("a8_residual_hidden"))
    assert data[1036] == 0xF0
    data[1036] = 0x7F
    return bytes(data)


def test_oracle_wraps_like_the_reference_where_a_residual_coefficient_runs_away(oracle):
    blob = _alpha_residual_with_a_runaway_coefficient()
    acodes, _, _, _, _, ae = oracle.decode_alpha(blob)
    assert ae == 0
    if oracle.have_reference():
        rc, plane = _alpha_with_reference(oracle, blob)
        assert rc == 0 and np.array_equal(np.asarray(plane).reshape(acodes.shape).astype(np.uint16), acodes)


@pytest.mark.gpu
def test_gpu_int32_residual_planes_of_up_to_12_bits_take_the_long_transform(oracle):
    """A residual frame with hidden bits keeps int32 coefficients; the reference transforms it with IDCT<.., QUAD> only beyond 12 bits
    (hidden bits included), with the LONG flavour otherwise (codestream/tables.cpp:1876-1891) -- the same numbers until a
    coefficient overflows 32 bits on the way, which the 64-bit flavour then does not reproduce."""
    blob = _alpha_residual_with_a_runaway_coefficient()
    acodes = oracle.decode_alpha(blob)[0]
    codes = oracle.decode_xt_status(blob)[0]
    d = api.Decoder(0)
    d.read(blob)
    assert np.array_equal(np.asarray(d.reconstruct()).reshape(-1).astype(np.uint16), np.asarray(codes).reshape(-1).astype(np.uint16))
    a = d.alpha_channel()
    assert a is not None and np.array_equal(np.asarray(a.reconstruct()).reshape(-1).astype(np.uint16), acodes.reshape(-1))
    d.close()
