"""Merging specification boxes in files that carry no residual codestream.

Tables::LTrafoTypeOf (codestream/tables.cpp:1994-2021) asks the box first: its LTRF decides the L transformation, whatever the
Adobe marker says; Zero / JPEG_LS / RCT and a box in a one-component frame are MALFORMED_STREAM; a free-form number nobody
defined is OBJECT_DOESNT_EXIST (colortrafo/colortransformerfactory.cpp:379-383).  The reference's encoder writes such files
for `-c` (SPEC{OCON, LTRF = identity}, tables.cpp:625-632) and for every grey scale picture that is not baseline (SPEC{OCON}).
The goldens refc_* / refspec_* (tests/golden/make_golden.py) pin the pictures; here: the variants of the box.
"""
import numpy as np
import pytest

from conftest import golden_jpeg
from libjpeg_amd import api


def patch(data, tag, val):
    i = data.index(tag) + 4
    return data[:i] + bytes([val]) + data[i + 1:]


def remove_adobe(data):
    i = data.index(b"\xff\xee\x00\x0eAdobe")
    return data[:i] + data[i + 16:]


def add_ltrf(data, v):
    """Append an LTRF box to the merging specification of a file whose SPEC box fits one APP11 segment."""
    i = data.index(b"SPEC")
    seg = data.rindex(b"\xff\xeb", 0, i)
    ln = (data[seg + 2] << 8) | data[seg + 3]
    body = bytearray(data[seg:seg + 2 + ln]) + b"\x00\x00\x00\x09LTRF" + bytes([v])
    body[2:4] = (ln + 9).to_bytes(2, "big")
    body[12:16] = (int.from_bytes(body[12:16], "big") + 9).to_bytes(4, "big")  # LBox of the SPEC box
    return data[:seg] + bytes(body) + data[seg + 2 + ln:]


def variants():
    base = golden_jpeg("refc_83x47_420")
    gray = golden_jpeg("refspec_70x40_gray")
    # name -> (stream, error the reference answers, YCbCr in force)
    out = {"base": (base, 0, 0), "ltrf_identity": (patch(base, b"LTRF", 0x10), 0, 0), "ltrf_ycbcr": (patch(base, b"LTRF", 0x20), 0, 1),
           "ltrf_ycbcr_no_adobe": (remove_adobe(patch(base, b"LTRF", 0x20)), 0, 1), "ltrf_identity_no_adobe": (remove_adobe(base), 0, 0),
           "ltrf_zero": (patch(base, b"LTRF", 0x00), -1038, None), "ltrf_jpegls": (patch(base, b"LTRF", 0x30), -1038, None),
           "ltrf_rct": (patch(base, b"LTRF", 0x40), -1038, None), "ltrf_freeform_5": (patch(base, b"LTRF", 0x50), -1031, None),
           "ltrf_freeform_15": (patch(base, b"LTRF", 0xF0), -1031, None), "gray": (gray, 0, 0), "gray_with_ltrf": (add_ltrf(gray, 0x10), -1038, None)}
    return out


VARIANTS = variants()


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_oracle_and_host_decoder_on_box_variants(oracle, name):
    data, err, ycc = VARIANTS[name]
    if oracle.have_reference():
        rpx, rerr = oracle.reference_decode_status(data)
        assert rerr == err, (name, rerr)
    opx, oerr, _ = oracle.decode_status(data)
    assert oerr == err
    if err == 0 and oracle.have_reference():
        assert np.array_equal(opx, rpx), name
    d = api.Decoder(None)
    try:
        try:
            f = d.read(data)
            assert err == 0 and f.ycbcr == ycc and f.xt == 0
            _, planes = oracle.decode_coefficients(data)
            for c in range(f.components):
                assert np.array_equal(d.coefficients(c), planes[c].astype(np.int16))
        except api.MijpegError as e:
            assert e.code == err, (name, e)
    finally:
        d.close()


def test_the_box_wins_over_the_adobe_marker(oracle):
    """Same coefficients, Adobe transform 0 in both, LTRF identity vs YCbCr: different pictures."""
    a, b = oracle.decode(VARIANTS["ltrf_identity"][0]), oracle.decode(VARIANTS["ltrf_ycbcr"][0])
    assert a.shape == b.shape and not np.array_equal(a, b)


# (output conversion byte, what the reference binary answers, bytes per sample of its picture)
OCON_VARIANTS = {0x00: (-1024, 0), 0x04: (-1024, 0), 0x03: (0, 1), 0x0A: (0, 1), 0x12: (0, 2), 0x82: (0, 2)}


@pytest.mark.parametrize("ocon", sorted(OCON_VARIANTS))
def test_output_conversions_beyond_the_plain_picture(oracle, ocon):
    """Round 4 declined all of these.  Without clamping no transformer exists (INVALID_PARAMETER, colortransformerfactory.cpp:
    698-725, 850-885); the output lookup indices and the lossless flag change nothing without a residual; extra range bits make
    the L chain stretch the picture to 8 + n bits (identity L tables scaled by ScaledTableOf).  The reference binary's verdict
    and picture where it is here, the oracle's, the product's host side."""
    want, sb = OCON_VARIANTS[ocon]
    blob = patch(VARIANTS["base"][0], b"OCON", ocon)
    codes, is_float, oerr = oracle.decode_xt_status(blob)
    assert oerr == want
    if oracle.have_reference():
        rpx, rerr = oracle.reference_decode_status(blob)
        assert rerr == want
        if want == 0:
            assert rpx.dtype.itemsize == sb and np.array_equal(rpx.astype(np.uint16), codes)
    d = api.Decoder(None)
    try:
        f = d.read(blob)
        assert want == 0 and f.sample_bytes == sb
        # (lookup indices and the lossless flag change nothing: the plain frame; more bits: the L chain, nothing merged)
        assert f.xt == (0 if ocon in (0x03, 0x0A) else 1) and (not f.xt or d.xt_params().no_residual == 1)
    except api.MijpegError as e:
        assert e.code == want
    d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ocon", sorted(k for k, v in OCON_VARIANTS.items() if v[0] == 0))
def test_gpu_output_conversions_beyond_the_plain_picture(oracle, ocon):
    blob = patch(VARIANTS["base"][0], b"OCON", ocon)
    codes, _, err = oracle.decode_xt_status(blob)
    assert err == 0
    d = api.Decoder(0)
    d.read(blob)
    out = d.reconstruct()
    d.close()
    assert out.dtype.itemsize == OCON_VARIANTS[ocon][1] and np.array_equal(out.astype(np.uint16).reshape(codes.shape), codes)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(k for k, v in VARIANTS.items() if v[1] == 0))
def test_gpu_pixels_of_box_variants(oracle, name):
    data = VARIANTS[name][0]
    exp = oracle.reference_decode(data) if oracle.have_reference() else oracle.decode(data)
    d = api.Decoder(0)
    try:
        d.read(data)
        assert np.array_equal(d.reconstruct(), exp), name
    finally:
        d.close()


def test_an_undefined_l_transformation_follows_the_adobe_marker(oracle):
    """A merging specification WITHOUT an LTRF box leaves the L transformation to the rule of plain JPEG (Tables::LTrafoTypeOf,
    codestream/tables.cpp:2021-2030): three components and no Adobe marker that says "none" -> YCbCr, else the identity -- with or
    without a residual codestream.  (Rounds 1-4 took YCbCr for granted beside a residual; found by a byte inserted into the box's
    type, tools/box_campaign.py.)"""
    import os

    import xt_craft
    from conftest import GOLDEN_DIR

    with open(os.path.join(GOLDEN_DIR, "xt_int8", "enc_444.jpg"), "rb") as f:
        base = f.read()
    no_ltrf = xt_craft.edit_spec(base, drop=(b"LTRF",))
    adobe_none = b"\xff\xee\x00\x0eAdobe\x00\x64\x00\x00\x00\x00\x00"
    with_adobe = no_ltrf[:2] + adobe_none + no_ltrf[2:]
    for blob, ycc in ((no_ltrf, 1), (with_adobe, 0)):
        codes, _, err = oracle.decode_xt_status(blob)
        assert err == 0
        if oracle.have_reference():
            rpx, rerr = oracle.reference_decode_status(blob)
            assert rerr == 0 and np.array_equal(rpx.astype(np.uint16), codes.reshape(rpx.shape))
        d = api.Decoder(None)
        f = d.read(blob)
        assert f.xt == 1 and d.xt_params().ltrafo_ycbcr == ycc and f.ycbcr == ycc
        d.close()
    a, b = oracle.decode_xt_status(no_ltrf)[0], oracle.decode_xt_status(with_adobe)[0]
    assert not np.array_equal(a, b)
