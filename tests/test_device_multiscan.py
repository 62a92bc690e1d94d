"""Progressive frames (SOF2) and JPEG XT frames with hidden refinement scans, entropy-decoded ON THE DEVICE (SURVEY 8 rows a1 / f3 /
f4; huffman_prog_kernel: one lane per restart interval of every scan -- DC / AC first passes with EOB runs,
codestream/sequentialscan.cpp:678-773, and successive approximation refinement, codestream/refinementscan.cpp:584-700,
223-232).  The coefficients the device leaves in HBM are compared with the host decoder's, plane by plane, and the pixels with
the oracle / the reference's goldens.  What the device path must decline (scans without restart markers beyond a few KiB:
AC refinement is serial by construction; the residual scan types; damaged streams) still decodes -- on the host."""
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from libjpeg_amd import api, synth

pytestmark = pytest.mark.gpu

SMALL = ["pilprog_200x130_422", "pilprog_70x40_gray", "pilprog_75x45_420", "refprog_120x88_420_qv", "refprog_64x64_444_dri5", "refprog_97x61_420",
         "refspec_70x41_gray_prog_dri", "refc_64x40_444_prog"]
XT_SMALL = ["xt_129x71_420_R2_rR3_dri3"]


def golden(name):
    with open(os.path.join(GOLDEN_DIR, name + ".jpg"), "rb") as f:
        return f.read()


@pytest.fixture(scope="module")
def dec():
    d = api.Decoder(0)
    yield d
    d.close()


@pytest.fixture(scope="module")
def hostdec():
    d = api.Decoder(None)
    yield d
    d.close()


def same_coefficients(dec, hostdec, data):
    """device entropy decode == host entropy decode, every plane of every frame; returns the info of the device read"""
    fi = hostdec.read(data, entropy="host")
    want = [hostdec.coefficients(c) for c in range(fi.components)]
    want_r = [hostdec.residual_coefficients(c) for c in range(hostdec.xt_params().residual.components)] if fi.xt else []
    info = dec.read(data, entropy="gpu")
    assert dec.entropy_used == "gpu"
    for c in range(info.components):
        got = dec.coefficients(c)
        assert got.dtype == want[c].dtype and np.array_equal(got, want[c]), f"component {c}"
    for c, w in enumerate(want_r):
        got = dec.residual_coefficients(c)
        assert got.dtype == w.dtype and np.array_equal(got, w), f"residual component {c}"
    assert list(info.range_max)[:info.components] == list(fi.range_max)[:fi.components]
    if fi.xt:
        assert list(dec.xt_params().residual.range_max)[:3] == list(hostdec.xt_params().residual.range_max)[:3]
    return info


@pytest.mark.parametrize("name", SMALL)
def test_progressive_goldens_on_the_device(oracle, dec, hostdec, name):
    data = golden(name)
    info = same_coefficients(dec, hostdec, data)
    assert info.progressive
    assert np.array_equal(dec.reconstruct(), oracle.decode(data))


@pytest.mark.parametrize("name", XT_SMALL)
def test_hidden_refinement_scans_on_the_device(oracle, dec, hostdec, name):
    data = golden(name)
    info = same_coefficients(dec, hostdec, data)
    assert info.xt and (dec.xt_params().hidden_bits > 0 or dec.xt_params().residual_hidden_bits > 0)
    codes, is_float, err = oracle.decode_xt_status(data)
    assert err == 0
    assert np.array_equal(dec.reconstruct(), codes)


ENCODER_VARIANTS = [
    (["-v", "-q", "85", "-s", "1x1,2x2,2x2", "-z", "4"], 333, 211),
    (["-v", "-qv", "-q", "90", "-z", "7"], 259, 131),
    (["-v", "-q", "60", "-s", "1x1,2x1,2x1", "-z", "1"], 97, 161),
    (["-v", "-h", "-q", "75", "-s", "1x1,2x2,2x2", "-z", "16"], 640, 360),  # optimised Huffman tables
    (["-v", "-q", "95", "-s", "1x1,1x2,1x2", "-z", "3"], 130, 97),
]


@pytest.mark.parametrize("args,w,h", ENCODER_VARIANTS)
def test_reference_encoded_progressive_with_restart_markers(oracle, dec, hostdec, args, w, h):
    if not oracle.have_reference():
        pytest.skip("needs the reference encoder")
    data = oracle.reference_encode(synth.synth_image(w, h, 41 + w), args)
    info = same_coefficients(dec, hostdec, data)
    assert info.progressive
    assert np.array_equal(dec.reconstruct(), oracle.decode(data))


XT_VARIANTS = [
    (["-r12", "-rR", "4", "-z", "8", "-s", "1x1,2x2,2x2"], 320, 200),      # config 5's hidden-bit variant with restart markers
    (["-r12", "-R", "2", "-rR", "3", "-z", "5", "-s", "1x1,2x2,2x2"], 257, 129),
    (["-r12", "-R", "4", "-z", "2"], 96, 64),                                # hidden bits in the legacy frame only, 4:4:4
    (["-r12", "-rv", "-z", "6", "-s", "1x1,2x2,2x2"], 200, 120),            # a progressive residual codestream
    (["-r12", "-v", "-rv", "-rR", "2", "-z", "4"], 150, 90),                 # both progressive, hidden bits on top
]


@pytest.mark.parametrize("extra,w,h", XT_VARIANTS)
def test_reference_encoded_xt_with_hidden_bits(oracle, dec, hostdec, extra, w, h):
    if not oracle.have_reference():
        pytest.skip("needs the reference encoder")
    data = oracle.reference_encode_hdr(synth.synth_hdr(w, h, 7 + w) * 4.0, ["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c"] + extra)
    same_coefficients(dec, hostdec, data)
    codes, is_float, err = oracle.decode_xt_status(data)
    assert err == 0
    assert np.array_equal(dec.reconstruct(), codes)


def test_what_the_device_path_declines_still_decodes(oracle, dec):
    """A progressive picture without restart markers whose scans are beyond a lane's reach: "gpu" says NOT_AVAILABLE, "prefer-gpu"
    ends up on the host, same pixels."""
    if not oracle.have_reference():
        pytest.skip("needs the reference encoder")
    data = oracle.reference_encode(synth.synth_image(1024, 768, 5), ["-v", "-q", "90"])
    with pytest.raises(api.MijpegError) as e:
        dec.read(data, entropy="gpu")
    assert e.value.code == api.ERR_NOT_AVAILABLE
    dec.read(data, entropy="prefer-gpu")
    assert dec.entropy_used == "host"
    assert np.array_equal(dec.reconstruct(), oracle.decode(data))


def test_damaged_progressive_streams_leave_the_device_path(oracle, dec, hostdec):
    """Bytes flipped inside the scans of a progressive stream with restart markers: whatever the device decoder makes of them, the
    answer is the host walk's (pixels or error code), never a wrong picture."""
    import damage

    data = golden("refprog_64x64_444_dri5")
    n = 0
    for kind, bad in damage.cases(data, 60, 20260930):
        try:
            hostdec.read(bad, entropy="host")
            want = None
        except api.MijpegError as e:
            want = e.code
        try:
            dec.read(bad, entropy="prefer-gpu")
            got = None
        except api.MijpegError as e:
            got = e.code
        assert got == want, kind
        if want is None:
            n += 1
            for c in range(dec.info.components):
                assert np.array_equal(dec.coefficients(c), hostdec.coefficients(c)), (kind, c)
    assert n >= 10


@pytest.mark.parametrize("sub,dri,prog", [("444", 2, False), ("420", 3, False), ("420", 4, True), ("422", 1, True)])
def test_12bit_frames_on_the_device(oracle, dec, hostdec, sub, dri, prog):
    """SOF1 / SOF2 frames of twelve bits (round 6: the device decoders took 8-bit frames only): same coefficient planes as the host
    decoder's, the 16-bit samples the oracle's; the goldens written by the reference with them."""
    data = synth.to_12bit(synth.synth_jpeg(328, 200, 31 + dri, 88, sub, dri, progressive=prog))
    info = same_coefficients(dec, hostdec, data)
    assert info.precision == 12 and bool(info.progressive) == prog
    assert np.array_equal(dec.reconstruct(), oracle.decode16(data))
    for name in ("p12_64x48_444", "p12_120x90_420_dri3"):
        data = golden(name)
        same_coefficients(dec, hostdec, data)
        assert np.array_equal(dec.reconstruct(), oracle.decode16(data))
