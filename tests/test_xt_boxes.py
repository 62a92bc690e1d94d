"""JPEG XT box framing under damage: what the reference's tables know of a box decides what the file is.

Box::ParseBoxMarker (boxes/box.cpp:93-200) reassembles a box from its APP11 segments and hands it to the tables only when the
payload bytes its segments ANNOUNCED reach the box length (codestream/tables.cpp:1187-1283); until then the box sits in the list
unparsed and nobody looks at it.  So:
  * a residual codestream (complete RESI box) and NO merging specification -- the SPEC segment's "JP" identifier damaged, or
    its length field too large for the box ever to fill up -- is read to the end, and the first request for pixels fails: no
    specification -> R transformation "zero" (tables.cpp:2070-2071), no transformer for that beside a residual frame
    (colortrafo/colortransformerfactory.cpp:277-291): -1024 "The combination of L and R transformation is non-standard ...";
  * a RESI box that is unknown or never complete leaves a merging specification without residual (tests/test_spec_boxes.py's
    subject): the legacy picture, where the specification asks for no more than that;
  * with neither, a plain JPEG;
  * damage INSIDE the residual codestream's entropy coded data that looks like a marker ends its scan early; the residual
    image's trailer then finds a marker, and Image::ParseResidualStream (codestream/image.cpp:1318-1331) hands the SAME frame
    back -- no second frame header is read as in the legacy codestream ("found a double frame header"), the frame goes on
    looking for scans and the picture decodes with what the residual scan had until then.
Found by a damage campaign over the one-component and integer JPEG XT goldens of round 4 (tests/golden/xt_grey, xt_int8,
xt_int16), which the campaigns of tests/test_xt_damaged.py (half-float streams) had not met.

Layers: oracle against the reference binary live (build container: return code, and pixels where a picture comes out); product
host decoder against the expectations below and the oracle (return codes, residual coefficient planes) on CPU; -m gpu: pixels.
"""
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

import damage
from conftest import GOLDEN_DIR, golden_jpeg
from libjpeg_amd import api

BASES = {
    "half_444_hidden": ("xt_64x48_444_R1_rR1.jpg", True),
    "half_420": ("xt_129x71_420.jpg", True),
    "grey_int": (os.path.join("xt_grey", "g8.jpg"), False),
    "grey_half": (os.path.join("xt_grey", "ghdr.jpg"), True),
    "grey_half_hidden": (os.path.join("xt_grey", "ghdr_R1_rR3.jpg"), True),
    "int8_444": (os.path.join("xt_int8", "enc_444.jpg"), False),
    "int16_420_r12": (os.path.join("xt_int16", "w420_r12.jpg"), False),
}


def base(name):
    with open(os.path.join(GOLDEN_DIR, BASES[name][0]), "rb") as f:
        return f.read()


def segments(data, box_type):
    """offsets of the APP11 segments (in front of the first scan) that carry a piece of this box"""
    out, i = [], 2
    while i + 4 <= len(data) and data[i] == 0xFF and data[i + 1] != 0xDA:
        ln = (data[i + 2] << 8) | data[i + 3]
        if data[i + 1] == 0xEB and data[i + 4:i + 6] == b"JP" and data[i + 16:i + 20] == box_type:
            out.append((i, ln))
        i += 2 + ln
    return out


def handmade(name):
    """kind -> (stream, what the reference answers: a code, 0 = a picture, "plain" = the legacy picture or a refusal)"""
    data = base(name)
    spec, resi = segments(data, b"SPEC"), segments(data, b"RESI")
    assert len(spec) == 1 and resi
    out = {}
    b = bytearray(data)
    b[spec[0][0] + 5] ^= 0xA7  # "JP" -> not a box: the segment is an unknown APP11 marker
    out["spec_tag"] = (bytes(b), -1024)
    b = bytearray(data)
    b[spec[0][0] + 12] = b[spec[0][0] + 13] = 0xFF  # LBox: the box never fills up
    out["spec_lbox"] = (bytes(b), -1024)
    b = bytearray(data)
    for off, _ in resi:
        b[off + 17] = 0  # "R\0SI": a box of unknown type, skipped
    out["resi_type"] = (bytes(b), "plain")
    b = bytearray(data)
    for off, _ in resi:  # one byte more announced than ever arrives (the same in every segment: box.cpp:155-157)
        lbox = int.from_bytes(b[off + 12:off + 16], "big") + 1
        b[off + 12:off + 16] = lbox.to_bytes(4, "big")
    out["resi_lbox"] = (bytes(b), "plain")
    b = bytearray(data)
    b[spec[0][0] + 5] ^= 0xA7
    for off, _ in resi:
        b[off + 17] = 0
    out["neither"] = (bytes(b), "plain")
    # fill bytes and a marker in the residual codestream's entropy coded data
    off, ln = resi[-1]
    seg = data[off:off + 2 + ln]
    sos = seg.find(b"\xff\xda") if len(resi) == 1 else 20
    a = off + sos + (2 + ln - sos) * 2 // 3
    b = bytearray(data)
    b[a:a + 4] = b"\xff\xff\xff\xb3"  # (FFB3: a frame header of the residual kind)
    out["resi_ff"] = (bytes(b), 0)
    # the residual codestream loses its last third (EOI included): segment length and box length say so, the framing is intact
    cut = (2 + ln - sos) // 3
    b = bytearray(data[:off + 2 + ln - cut] + data[off + 2 + ln:])
    b[off + 2:off + 4] = (ln - cut).to_bytes(2, "big")
    for o, _ in resi:
        lbox = int.from_bytes(b[o + 12:o + 16], "big") - cut
        b[o + 12:o + 16] = lbox.to_bytes(4, "big")
    out["resi_cut"] = (bytes(b), 0)
    # a refinement box whose scan header is gone: skipped with a warning, the frame turns to the next box
    # (Frame::StartParseScan, marker/frame.cpp:805-822, 857-860)
    for kind, tbox in (("fine_without_scan", b"FINE"), ("rfin_without_scan", b"RFIN")):
        boxes = segments(data, tbox)
        if boxes:
            o, l = boxes[0]
            at = data.index(b"\xff\xda", o, o + 2 + l)
            b = bytearray(data)
            b[at + 1] = 0x01
            out[kind] = (bytes(b), 0)
    return out


CASES = {(n, k): v for n in BASES for k, v in handmade(n).items()}


def reference_status(oracle, data, is_float):
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
        src, dst = os.path.join(d, "in.jpg"), os.path.join(d, "out.pfm" if is_float else "out.ppm")
        with open(src, "wb") as f:
            f.write(data)
        r = subprocess.run([oracle.REF_BIN, src, dst], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=30)
        assert r.returncode >= 0, "the reference crashed"
        m = re.search(rb"failed - error (-?\d+)", r.stderr)
        if m:
            return None, int(m.group(1))
        with open(dst, "rb") as f:
            magic = f.read(2)
        return (oracle.read_pfm_reference(dst) if magic in (b"PF", b"Pf") else oracle.read_pnm_any(dst)), 0


def oracle_picture(oracle, data):
    """-> (pixels or None, code): the XT restatement, or -- a file the reference takes for a legacy picture -- the plain one"""
    codes, is_float, oerr = oracle.decode_xt_status(data)
    if codes is not None:
        return (oracle.half_codes_to_float(codes) if is_float else codes), 0
    if oerr is not None:
        return None, oerr
    try:
        return oracle.decode(data), 0
    except ValueError:
        return None, None


# ------------------------------------------------------------------------------------------------ oracle
@pytest.mark.parametrize("name", sorted(BASES))
def test_oracle_against_live_reference(oracle, name):
    if not oracle.have_reference():
        pytest.skip("oracle/_ref/jpeg not built")
    compared = 0
    for (n, kind), (blob, expect) in CASES.items():
        if n != name:
            continue
        rpx, rerr = reference_status(oracle, blob, BASES[name][1])
        assert rerr == (expect if isinstance(expect, int) else 0), (name, kind, rerr)
        opx, oerr = oracle_picture(oracle, blob)
        # (the legacy picture through a specification that asks for more than the plain picture -- L tables, float output -- was
        # outside the restatement's subset until round 5: tests/test_xt_lonly.py)
        assert oerr == rerr, (name, kind, oerr, rerr)
        if rerr == 0:
            opx = opx.reshape(rpx.shape) if opx.size == rpx.size else opx
            assert opx.shape == rpx.shape and np.array_equal(opx.astype(rpx.dtype), rpx), (name, kind)
        compared += 1
    assert compared >= 4


def test_damaged_residual_scan_changes_the_picture_not_the_verdict(oracle):
    for name in BASES:
        for kind in ("resi_ff", "resi_cut"):
            blob, _ = CASES[(name, kind)]
            a, _, ea = oracle.decode_xt_status(base(name))
            b, _, eb = oracle.decode_xt_status(blob)
            assert ea == 0 and eb == 0 and a.shape == b.shape and not np.array_equal(a, b), (name, kind)


# ------------------------------------------------------------------------------------------------ product, host side
def test_host_decoder_verdicts(oracle):
    d = api.Decoder(None)
    declined = 0
    for (name, kind), (blob, expect) in sorted(CASES.items()):
        try:
            f = d.read(blob)
            perr = 0
        except api.MijpegError as e:
            perr = e.code
        if expect == "plain":
            # the legacy picture: plain, or -- where the specification wants tables / more bits / float output -- through the L
            # chain with nothing merged (an XT frame without residual planes)
            assert perr == 0, (name, kind, perr)
            assert not f.xt or (d.xt_params().no_residual == 1 and d.xt_params().residual.components == 0 and d.xt_params().residual_hidden_bits == 0), (name, kind)
            declined += perr != 0
            _, oerr = oracle_picture(oracle, blob)
            assert oerr == 0, (name, kind, oerr, perr)
        else:
            assert perr == expect, (name, kind, perr)
            _, _, oerr = oracle.decode_xt_status(blob)
            assert oerr == expect, (name, kind, oerr)
    d.close()
    assert declined == 0


def test_legacy_planes_where_the_residual_box_is_not_known(oracle):
    """... and the coefficients of the legacy picture: refinement boxes are read behind the last visible scan whether or not a
    specification says how many bits hide in them (marker/frame.cpp:1063-1070) -- `half_444_hidden` has one, and without
    specification its scan refines a bit the visible scans have written already."""
    def legacy_planes(blob):
        # (a specification that is still there and names hidden bits moves the visible scans up: the merge's view of the frame)
        try:
            return oracle.decode_xt_planes(blob)
        except ValueError:
            return oracle.decode_coefficients(blob)

    d = api.Decoder(None)
    n = refined = 0
    for (name, kind), (blob, expect) in sorted(CASES.items()):
        if expect != "plain":
            continue
        f = d.read(blob)
        info, planes = legacy_planes(blob)
        for c in range(info.ncomp):
            assert np.array_equal(d.coefficients(c).astype(np.int32), planes[c]), (name, kind, c)
        if name == "half_444_hidden":
            fine = segments(blob, b"FINE")
            without = bytearray(blob)
            for o, _ in fine:
                without[o + 5] ^= 0xA7
            visible = legacy_planes(bytes(without))[1]
            refined += any(not np.array_equal(visible[c], planes[c]) for c in range(info.ncomp))
        n += 1
    d.close()
    assert n >= 6 and refined >= 1


def test_header_only_read_reports_the_missing_specification():
    d = api.Decoder(None)
    for name in BASES:
        with pytest.raises(api.MijpegError) as e:
            d.read_header(CASES[(name, "spec_tag")][0])
        assert e.value.code == -1024 and "non-standard" in str(e.value)
    d.close()


def test_errors_of_either_codestream_come_first(oracle):
    """The reference reads the whole file before it builds the transformer: a legacy or residual scan that does not decode is
    what it reports, -1024 only behind both."""
    if not oracle.have_reference():
        pytest.skip("oracle/_ref/jpeg not built")
    d = api.Decoder(None)
    seen = set()
    for name in ("int8_444", "half_420", "int16_420_r12"):
        blob, _ = CASES[(name, "spec_tag")]
        for where in ("entropy", "any"):
            for kind, hurt in damage.cases(blob, 25, 9100 + len(name), where):
                _, rerr = reference_status(oracle, hurt, BASES[name][1])
                _, _, oerr = oracle.decode_xt_status(hurt)
                try:
                    d.read(hurt)
                    perr = 0
                except api.MijpegError as e:
                    perr = e.code
                if oerr is None:  # (a second hit made a plain JPEG of it)
                    continue
                assert oerr == rerr, (name, kind, oerr, rerr)
                assert perr == oerr or (perr == -1034 and oerr == 0), (name, kind, perr, oerr)
                seen.add(rerr)
    d.close()
    assert -1024 in seen and len(seen) >= 2, seen


def test_residual_planes_of_a_scan_that_ends_in_a_marker(oracle):
    """The product's host decoder walks the damaged residual codestream like the restatement does: the same coefficients."""
    d = api.Decoder(None)
    n = 0
    for name in BASES:
        for kind in ("resi_ff", "resi_cut"):
            blob, _ = CASES[(name, kind)]
            f = d.read(blob)
            assert f.xt
            x = d.xt_params()
            if x.residual_hidden_bits:
                continue  # (the restatement's entry for this test decodes the visible scans only)
            info, planes = oracle.decode_residual_coefficients(blob)
            intact = oracle.decode_residual_coefficients(base(name))[1]
            assert any(not np.array_equal(planes[c], intact[c]) for c in range(info.ncomp)), (name, kind)
            for c in range(info.ncomp):
                assert np.array_equal(d.residual_coefficients(c).astype(np.int32), planes[c]), (name, kind, c)
            n += 1
    d.close()
    assert n >= 8


# ------------------------------------------------------------------------------------------------ the transformer's refusals
# tests/golden/xt_int8 holds three streams whose merging specification names tables that do not exist or do not fit: the colour
# transformer refuses them when the first request builds it (Tables::ColorTrafoOf, codestream/tables.cpp:1517-1555).  The command
# line has read the whole file by then (cmd/reconstruct.cpp:119-121): what stops either codestream is reported instead, and a
# table of the residual's side is not even looked up when the legacy codestream has no EOI (no residual frame to merge).
LATE = {"a_q_table_missing": -1031, "a_q_is_tone_box": -1031, "a_r2_linear_negative_slope": -1024,
        # made here: transformations that name a matrix nobody defined -- the residual's is looked up beside a residual frame only,
        # the base one always (colortransformerfactory.cpp:355-400, 528-566)
        "craft_r_matrix_missing": -1031, "craft_l_matrix_missing": -1031}
ALWAYS = ("craft_l_matrix_missing",)  # refusals that do not depend on a residual frame


def late_cases(name):
    """kind -> (stream, the reference's answer; 0: a picture -- the legacy one through the L tables)"""
    if name.startswith("craft_"):
        import xt_craft as X
        with open(os.path.join(GOLDEN_DIR, "xt_int8", "enc_444.jpg"), "rb") as f:
            data = f.read()
        t = b"RTRF" if name == "craft_r_matrix_missing" else b"LTRF"
        data = X.edit_spec(data, X.subbox(t, b"\x70"), drop=(t,))
    else:
        with open(os.path.join(GOLDEN_DIR, "xt_int8", name + ".jpg"), "rb") as f:
            data = f.read()
    es = damage.entropy_start(data)
    out = {"intact": (data, LATE[name])}
    b = bytearray(data)
    b[es + 100:es + 104] = bytes(4)
    b[es + 140:es + 142] = b"\xff\xc4"  # a DHT marker in the legacy scan: "undefined Huffman table type"
    out["legacy_scan"] = (bytes(b), -1038)
    off, ln = segments(data, b"RESI")[-1]
    b = bytearray(data)
    b[off + ln - 120:off + ln - 114] = b"\x55" * 6  # the residual scan runs out of sync
    out["residual_scan"] = (bytes(b), -1038)
    out["no_eoi"] = (data[:-2], LATE[name] if name in ALWAYS else 0)
    out["cut"] = (data[:es + 150], LATE[name] if name in ALWAYS else 0)
    return out


@pytest.mark.parametrize("name", sorted(LATE))
def test_transformer_refusals_come_behind_both_codestreams(oracle, name):
    d = api.Decoder(None)
    for kind, (blob, expect) in late_cases(name).items():
        rpx = None
        if oracle.have_reference():
            rpx, rerr = reference_status(oracle, blob, False)
            assert rerr == expect, (name, kind, rerr)
        codes, _, oerr = oracle.decode_xt_status(blob)
        try:
            f = d.read(blob)
            perr = 0
        except api.MijpegError as e:
            perr = e.code
        assert oerr == expect and perr == expect, (name, kind, oerr, perr)
        if expect == 0:
            # the legacy picture through the L chain alone (round 4: outside the restatement, declined by the product)
            # (a specification that asks for no more than the plain picture gives the plain frame)
            if f.xt:
                x = d.xt_params()
                assert x.no_residual == 1 and x.residual.components == 0 and x.residual_hidden_bits == 0, (name, kind)
            if rpx is not None:
                assert np.array_equal(codes.reshape(rpx.shape), rpx), (name, kind)
    d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(LATE))
def test_gpu_legacy_picture_without_an_eoi_beside_refused_residual_tables(oracle, name):
    dec = api.Decoder(0)
    n = 0
    for kind, (blob, expect) in late_cases(name).items():
        if expect != 0:
            continue
        codes, _, oerr = oracle.decode_xt_status(blob)
        assert oerr == 0
        dec.read(blob)
        out = dec.reconstruct()
        assert np.array_equal(out.reshape(codes.shape), codes.astype(out.dtype)), (name, kind)
        n += 1
    dec.close()
    assert n == (0 if name in ALWAYS else 2)


# ------------------------------------------------------------------------------------------------ the specification's form
# What is wrong with the FORM of a merging specification -- sub-box framing, a kind of box twice, payload sizes, reserved bits,
# two curves or matrices for one index -- is found where the SPEC box completes (SuperBox::ParseBoxContent, boxes/superbox.cpp:
# 93-206; MergingSpecBox::CreateBox / AcknowledgeBox, boxes/mergingspecbox.cpp:108-256; the sub-boxes' ParseBoxContent), with
# or without a residual codestream: -1038 in front of everything that follows in the file.
def malformed_specifications(data):
    import struct
    import xt_craft as X
    curve = X.curv(9, "linear", 0, (0.0, 1.0, 0, 0))
    matrix = bytes([0x5D]) + struct.pack(">9h", 8192, 0, 0, 0, 8192, 0, 0, 0, 8192)
    raw = lambda lbox, tbox, payload: struct.pack(">L", lbox) + tbox + payload  # noqa: E731
    return {
        "ocon_twice": X.edit_spec(data, X.subbox(b"OCON", b"\x02\x00\x00")),
        "ocon_size": X.edit_spec(data, X.subbox(b"OCON", b"\x02\x00"), drop=(b"OCON",)),
        "ocon_lookup_bits": X.edit_spec(data, X.subbox(b"OCON", b"\x02\x00\x10"), drop=(b"OCON",)),
        "ocon_nine_extra_bits": X.edit_spec(data, X.subbox(b"OCON", b"\x92\x00\x00"), drop=(b"OCON",)),
        "ltrf_reserved": X.edit_spec(data, X.subbox(b"LTRF", b"\x21"), drop=(b"LTRF",)),
        "ltrf_size": X.edit_spec(data, X.subbox(b"LTRF", b"\x20\x00"), drop=(b"LTRF",)),
        "ctrf_twice": X.edit_spec(data, X.subbox(b"CTRF", b"\x10") + X.subbox(b"CTRF", b"\x10")),
        "lpts_size": X.edit_spec(data, X.subbox(b"LPTS", b"\x00\x00\x00"), drop=(b"LPTS",)),
        "spts_size": X.edit_spec(data, X.subbox(b"SPTS", b"\x00")),
        "rdct_type": X.edit_spec(data, X.subbox(b"RDCT", b"\x10")),
        "rdct_noise_without_bypass": X.edit_spec(data, X.subbox(b"RDCT", b"\x01")),
        "ldct_size": X.edit_spec(data, X.subbox(b"LDCT", b"\x00\x00")),
        "rspc_size": X.edit_spec(data, X.subbox(b"RSPC", b"\x00\x00"), drop=(b"RSPC",)),
        "rspc_five": X.edit_spec(data, X.subbox(b"RSPC", b"\x50"), drop=(b"RSPC",)),
        "two_curves_one_index": X.edit_spec(data, X.subbox(b"CURV", curve) + X.subbox(b"CURV", curve)),
        "curve_type": X.edit_spec(data, X.subbox(b"CURV", bytes([0x93]) + curve[1:])),
        "two_matrices_one_index": X.edit_spec(data, X.subbox(b"MTRX", matrix) + X.subbox(b"MTRX", matrix)),
        "matrix_index": X.edit_spec(data, X.subbox(b"MTRX", bytes([0x4D]) + matrix[1:])),
        "matrix_fraction": X.edit_spec(data, X.subbox(b"MTRX", bytes([0x5C]) + matrix[1:])),
        "alpha_composition": X.edit_spec(data, X.subbox(b"AMUL", b"\x00\x00\x00\x00\x00\x00\x00")),
        "child_length_zero": X.edit_spec(data, raw(0, b"XXXX", b"")),
        "child_length_four": X.edit_spec(data, raw(4, b"XXXX", b"")),
        "child_beyond_the_box": X.edit_spec(data, raw(64, b"XXXX", b"\x00")),
        "child_header_cut": X.edit_spec(data, b"\x00\x00\x00"),
        # ... and what is fine: an unknown child, a second curve for another index
        "unknown_child": X.edit_spec(data, X.subbox(b"XXXX", b"\x01\x02\x03")),
        "two_curves": X.edit_spec(data, X.subbox(b"CURV", curve) + X.subbox(b"CURV", X.curv(10, "linear", 0, (0.0, 1.0, 0, 0)))),
    }


SPEC_BASES = {"with_residual": os.path.join("xt_int8", "enc_444.jpg"), "without_residual": "refc_83x47_422.jpg"}
FINE = ("unknown_child", "two_curves")


@pytest.mark.parametrize("which", sorted(SPEC_BASES))
def test_form_of_the_merging_specification(oracle, which):
    with open(os.path.join(GOLDEN_DIR, SPEC_BASES[which]), "rb") as f:
        data = f.read()
    d = api.Decoder(None)
    for kind, blob in malformed_specifications(data).items():
        expect = 0 if kind in FINE else -1038
        if oracle.have_reference():
            _, rerr = reference_status(oracle, blob, False)
            assert rerr == expect, (which, kind, rerr)
        info = oracle.OjInfo()
        import ctypes as C
        rc = oracle.lib().oj_read_info(blob, len(blob), C.byref(info))
        assert (info.ref_error if rc else 0) == expect, (which, kind, rc, info.ref_error)
        if which == "with_residual":
            assert oracle.decode_xt_status(blob)[2] == expect, (which, kind)
        try:
            d.read(blob)
            perr = 0
        except api.MijpegError as e:
            perr = e.code
        assert perr == expect, (which, kind, perr)
        if expect:  # a header-only read says so as well
            with pytest.raises(api.MijpegError) as e:
                d.read_header(blob)
            assert e.value.code == expect
    d.close()


# ------------------------------------------------------------------------------------------------ product, pixels
@pytest.mark.gpu
def test_gpu_pixels_behind_a_damaged_residual_scan(oracle):
    dec = api.Decoder(0)
    for name in BASES:
        for kind in ("resi_ff", "resi_cut", "fine_without_scan", "rfin_without_scan"):
            if (name, kind) not in CASES:
                continue
            blob, _ = CASES[(name, kind)]
            codes, _, oerr = oracle.decode_xt_status(blob)
            assert oerr == 0
            dec.read(blob)
            out = dec.reconstruct()
            assert out.shape == codes.shape and np.array_equal(out, codes.astype(out.dtype)), (name, kind)
    dec.close()


@pytest.mark.gpu
def test_gpu_legacy_picture_where_the_residual_box_is_not_known(oracle):
    dec = api.Decoder(0)
    n = 0
    for (name, kind), (blob, expect) in sorted(CASES.items()):
        if expect != "plain":
            continue
        f = dec.read(blob)  # (round 4 declined the ones whose specification wants L tables / float output)
        codes, is_float, oerr = oracle.decode_xt_status(blob)
        exp = codes if codes is not None else oracle.decode(blob)
        out = dec.reconstruct()
        assert np.array_equal(out.reshape(exp.shape), exp.astype(out.dtype)), (name, kind)
        n += 1
    dec.close()
    assert n >= 20


@pytest.mark.gpu
def test_gpu_children_the_specification_does_not_know_change_nothing(oracle):
    dec = api.Decoder(0)
    for which, path in SPEC_BASES.items():
        with open(os.path.join(GOLDEN_DIR, path), "rb") as f:
            data = f.read()
        dec.read(data)
        want = dec.reconstruct().copy()
        variants = malformed_specifications(data)
        for kind in FINE:
            dec.read(variants[kind])
            assert np.array_equal(dec.reconstruct(), want), (which, kind)
    dec.close()


# ------------------------------------------------------------------------------------------------ tables beyond the range
# tests/golden/xt_boxes/r2_curve_overflows.jpg: xt_int8/a_r2_exponential.jpg with the curve's second parameter at 1024 (one
# byte of its CURV box): most entries of the R2 table leave 32 bits, LONG(double) makes 0x80000000 of them
# (boxes/parametrictonemappingbox.cpp:411-421), and the merge adds that in LONG variables -- it wraps, and 410 samples saturate
# at the top where sums in 64 bits would saturate at the bottom (colortrafo/ycbcrtrafo.cpp:776-789, 868-879).  The .bin holds
# the reference decoder's samples.
def overflowing_curve():
    with open(os.path.join(GOLDEN_DIR, "xt_boxes", "r2_curve_overflows.jpg"), "rb") as f:
        data = f.read()
    want = np.fromfile(os.path.join(GOLDEN_DIR, "xt_boxes", "r2_curve_overflows.bin"), np.uint8).reshape(32, 48, 3)
    return data, want


def test_oracle_merges_in_32_bits(oracle):
    data, want = overflowing_curve()
    codes, is_float, err = oracle.decode_xt_status(data)
    assert err == 0 and not is_float and np.array_equal(codes.astype(np.uint8), want) and codes.max() <= 255
    assert np.count_nonzero(want == 255) >= 400
    if oracle.have_reference():
        rpx, rerr = reference_status(oracle, data, False)
        assert rerr == 0 and np.array_equal(rpx, want)


@pytest.mark.gpu
def test_gpu_merges_in_32_bits(oracle):
    data, want = overflowing_curve()
    dec = api.Decoder(0)
    dec.read(data)
    out = dec.reconstruct()
    dec.close()
    assert out.shape == want.shape and np.array_equal(out, want)


# ------------------------------------------------------------------------------------------------ the file's own tables
# TONE, CURV and MTRX boxes of the file's list parse where they complete, and a table index / matrix id may be taken once
# (codestream/tables.cpp:1247-1266) -- in a plain JPEG as well, where nobody would ever look the table up.
def file_level_boxes():
    import struct
    import xt_craft as X
    tone = lambda idx, n=256: bytes([idx << 4 | 8]) + b"".join(struct.pack(">H", (i * 255) & 0xFFFF) for i in range(n))  # noqa: E731
    curve = lambda idx, ty=5: bytes([(idx << 4) | ty, 0]) + struct.pack(">4f", 0.0, 1.0, 0.0, 0.0)  # noqa: E731
    matrix = lambda i, t=13: bytes([(i << 4) | t]) + struct.pack(">9h", 8192, 0, 0, 0, 8192, 0, 0, 0, 8192)  # noqa: E731
    return {
        "tone": (X.app11(b"TONE", tone(9)), 0),
        "tone_even_size": (X.app11(b"TONE", tone(9) + b"\x00"), -1038),
        "tone_short": (X.app11(b"TONE", tone(9, 128)), -1038),
        "tone_not_a_power_of_two": (X.app11(b"TONE", tone(9, 300)), -1038),
        "two_tones_one_index": (X.app11(b"TONE", tone(9)) + X.app11(b"TONE", tone(9), 2), -1038),
        "two_tones": (X.app11(b"TONE", tone(9)) + X.app11(b"TONE", tone(10), 2), 0),
        "curve_then_tone_one_index": (X.app11(b"CURV", curve(9)) + X.app11(b"TONE", tone(9)), -1038),
        "tone_then_curve_one_index": (X.app11(b"TONE", tone(9)) + X.app11(b"CURV", curve(9)), 0),
        "curve_type": (X.app11(b"CURV", curve(9, 3)), -1038),
        "curve_size": (X.app11(b"CURV", curve(9) + b"\x00"), -1038),
        "matrix": (X.app11(b"MTRX", matrix(7)), 0),
        "two_matrices_one_id": (X.app11(b"MTRX", matrix(7)) + X.app11(b"MTRX", matrix(7), 2), -1038),
        "matrix_id": (X.app11(b"MTRX", matrix(4)), -1038),
        "matrix_fraction": (X.app11(b"MTRX", matrix(7, 12)), -1038),
    }


@pytest.mark.parametrize("which", ["plain", "xt"])
def test_tables_of_the_files_own_list(oracle, which):
    import ctypes as C
    import xt_craft as X
    if which == "plain":
        data = golden_jpeg("ref_16x16_420")
    else:
        with open(os.path.join(GOLDEN_DIR, "xt_int8", "enc_444.jpg"), "rb") as f:
            data = f.read()
    d = api.Decoder(None)
    for kind, (boxes, expect) in file_level_boxes().items():
        blob = data[:2] + boxes + data[2:] if which == "plain" else X.edit_spec(data, new_boxes=boxes)
        if oracle.have_reference():
            _, rerr = reference_status(oracle, blob, False)
            assert rerr == expect, (which, kind, rerr)
        info = oracle.OjInfo()
        rc = oracle.lib().oj_read_info(blob, len(blob), C.byref(info))
        assert (info.ref_error if rc else 0) == expect, (which, kind, rc, info.ref_error)
        try:
            d.read(blob)
            perr = 0
        except api.MijpegError as e:
            perr = e.code
        assert perr == expect, (which, kind, perr)
    d.close()


def test_refusal_beside_a_residual_scan_only_the_sequential_walk_rejects(oracle):
    """tests/golden/xt_boxes/refused_spec_damaged_residual.jpg (tools/box_campaign.py xt 9: xt_int8/a_q_is_tone_box.jpg with two
    bytes of its residual codestream hit): the residual scan decodes interval by interval without complaint and only the
    reference's sequential walk runs out of sync -- the verdict in front of the transformer's -1031 has to take that walk."""
    with open(os.path.join(GOLDEN_DIR, "xt_boxes", "refused_spec_damaged_residual.jpg"), "rb") as f:
        blob = f.read()
    if oracle.have_reference():
        assert reference_status(oracle, blob, False)[1] == -1038
    assert oracle.decode_xt_status(blob)[2] == -1038
    d = api.Decoder(None)
    with pytest.raises(api.MijpegError) as e:
        d.read(blob)
    d.close()
    assert e.value.code == -1038 and "out of sync" in str(e.value)


def _resegment(data, types=(b"RESI", b"RFIN", b"FINE"), piece=3000):
    """The boxes of `types` cut into APP11 segments of `piece` payload bytes (same instance number, sequence numbers 1, 2, ...:
    Box::ParseBoxMarker, boxes/box.cpp:93-200, puts them together again) -- what the reference's encoder does from 64 KiB on."""
    import struct
    out = bytearray(data[:2])
    p = 2
    while p < len(data) - 4 and data[p] == 0xFF:
        m = data[p + 1]
        ln = struct.unpack(">H", data[p + 2:p + 4])[0]
        seg = data[p:p + 2 + ln]
        if m == 0xEB and seg[4:6] == b"JP" and seg[16:20] in types and ln - 18 > piece:
            en, lbox, tbox, payload = seg[6:8], seg[12:16], seg[16:20], seg[20:]
            assert struct.unpack(">I", seg[8:12])[0] == 1
            for k, a in enumerate(range(0, len(payload), piece)):
                part = payload[a:a + piece]
                out += b"\xff\xeb" + struct.pack(">H", 18 + len(part)) + b"JP" + en + struct.pack(">I", k + 1) + lbox + tbox + part
        else:
            out += seg
        if m == 0xDA:
            out += data[p + 2 + ln:]
            break
        p += 2 + ln
    return bytes(out)


def test_codestream_boxes_of_many_segments(oracle):
    """A codestream box of many APP11 segments is not copied segment by segment: the walk notes its pieces and the pool copies them
    when it is through (XtBox::pieces, HostDecoder::materialize_boxes), into a store a box of the last parse left behind where
    one fits.  Same planes as the file with one segment per box, same verdicts where the file ends inside a segment (the reference
    fills up with zeros, io/decoderstream.cpp:136-158), and nothing of one read shows in the next on the same object."""
    data = golden_jpeg("xt_200x120_420_R3_rR4")
    multi = _resegment(data, piece=700)
    assert multi != data and multi.count(b"RESI") > 3
    want = oracle.decode_xt_status(data)
    got = oracle.decode_xt_status(multi)
    assert want[2] == 0 and got[2] == 0 and np.array_equal(want[0], got[0])
    if oracle.have_reference():
        assert np.array_equal(oracle.reference_decode_hdr(data), oracle.reference_decode_hdr(multi))
    one = api.Decoder(None)
    f = one.read(data)
    planes = [one.coefficients(c).copy() for c in range(f.components)] + [one.residual_coefficients(c).copy() for c in range(f.components)]
    one.close()
    # cuts inside the second segment of the residual codestream's box, inside a refinement box, and right behind a segment header
    r = [m.start() for m in re.finditer(b"RESI", multi)]
    cuts = [multi[:r[1] + 700], multi[:r[2] + 4], multi[:multi.index(b"RFIN") + 1500], multi[:r[-1] + 10]]
    verdicts = []
    for blob in cuts:
        oerr = oracle.decode_xt_status(blob)[2]
        if oracle.have_reference():
            assert _reference_error(oracle, blob) == oerr
        verdicts.append(oerr)
    d = api.Decoder(None)  # ONE object through all of them, the whole file in between
    for rep in range(2):
        for blob, oerr in zip(cuts, verdicts):
            f = d.read(multi)
            now = [d.coefficients(c) for c in range(f.components)] + [d.residual_coefficients(c) for c in range(f.components)]
            assert all(np.array_equal(a, b) for a, b in zip(planes, now))
            if oerr == 0:
                d.read(blob)
            else:
                with pytest.raises(api.MijpegError) as e:
                    d.read(blob)
                assert e.value.code == oerr
    d.close()


def _reference_error(oracle, blob):
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as t:
        src, dst = os.path.join(t, "in.jpg"), os.path.join(t, "out.pfm")
        with open(src, "wb") as f:
            f.write(blob)
        r = subprocess.run([oracle.REF_BIN, src, dst], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=20)
        m = re.search(rb"failed - error (-?\d+)", r.stderr)
        return int(m.group(1)) if m else 0
