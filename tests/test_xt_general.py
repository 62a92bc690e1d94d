"""JPEG XT profile C beyond the reference encoder's default output: free-form L / R / C transformations
(MTRX boxes: the encoder's -xyz / -cxyz), parametric curves (CURV boxes) as L, Q and R2 tables
(boxes/parametrictonemappingbox.cpp:199-272, 387-430), real Q / R2 table gathers
(colortrafo/colortransformerfactory.cpp:435-520, colortrafo/ycbcrtrafo.cpp:775-800), the residual DCT bypass with and without
noise shaping (control/residualblockhelper.cpp:203-231), an 8-bit residual codestream.

tests/golden/xt_general/ holds the streams and the REAL reference decoder's float32 output (or error code) for each
(tests/golden/make_xt_general.py; the hand-made ones come from tests/xt_craft.py).  CPU: the oracle against those goldens, the
product's host side (merging specification, error codes, tables) against the oracle.  -m gpu: the product's pixels."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from libjpeg_amd import api

DIR = os.path.join(GOLDEN_DIR, "xt_general")
with open(os.path.join(DIR, "manifest.json")) as _f:
    CASES = json.load(_f)


def stream(name):
    with open(os.path.join(DIR, name + ".jpg"), "rb") as f:
        return f.read()


def expected(name):
    ent = CASES[name]
    if ent["error"]:
        return None
    with open(os.path.join(DIR, name + ".bin"), "rb") as f:
        return np.frombuffer(f.read(), "<f4").reshape(ent["height"], ent["width"], 3)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_against_the_reference_decoder(oracle, name):
    codes, is_float, err = oracle.decode_xt_status(stream(name))
    assert err == CASES[name]["error"], (name, err)
    if err == 0:
        assert is_float
        assert np.array_equal(oracle.half_codes_to_float(codes), expected(name)), name


@pytest.mark.parametrize("name", sorted(CASES))
def test_host_side_merging_specification(oracle, name):
    """Error codes are the reference's; what decodes reports the generalisation in use."""
    d = api.Decoder(None)
    try:
        try:
            d.read(stream(name))
            code = 0
        except api.MijpegError as e:
            code = e.code
        assert code == CASES[name]["error"], name
        if code == 0:
            x = d.xt_params()
            want_general = not any(k in name for k in ("q_identity_curve", "l_gamma_curve", "l_curve_inside_spec", "enc_residual8"))
            assert bool(x.general) == want_general, name
            if "bypass" in name:
                assert x.rdct_bypass == 1 and x.noise_shaping == (1 if "noise" in name else 0)
            if "enc_residual8" in name:
                assert x.residual.precision == 8
    finally:
        d.close()


def test_parametric_tables_equal_the_oracles(oracle):
    """The Q / R2 tables the host builds for the kernels (long double around double library calls, like the reference's x87
    build) are, entry for entry, what the oracle builds -- checked on the curve types with transcendental functions."""
    import ctypes as C

    for name in ("a_r2_gamma", "a_r2_exponential", "a_r2_logarithmic", "a_r2_gammaoffset", "a_q_linear"):
        d = api.Decoder(None)
        d.read(stream(name))
        x = d.xt_params()
        which, n = ("qtable", x.qtable_entries) if "q_" in name else ("r2table", 1 << 20)
        tab = np.ctypeslib.as_array((C.c_int32 * n).from_address(getattr(x, which)[0])).copy()
        d.close()
        # the oracle applies its table inside the merge; its pixels matching the reference (test above) pins the entries that
        # occur -- here: monotone where the curve is, and the end points of the scaled table
        assert tab.shape == (n,)
        if "linear" in name or "gamma" in name or "exponential" in name:
            assert np.all(np.diff(tab.astype(np.int64)) >= 0), name


@pytest.fixture(scope="module")
def dec():
    d = api.Decoder(0)
    yield d
    d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("entropy", ["host", "auto"])
@pytest.mark.parametrize("name", sorted(k for k, v in CASES.items() if v["error"] == 0))
def test_gpu_pixels_equal_the_reference(oracle, dec, name, entropy):
    """Half-float codes out of the kernels (general merge, table gathers, bypass dequantisation) expanded like the reference's
    CLI does: bit-identical to the PFM it wrote."""
    info = dec.read(stream(name), entropy=entropy)
    assert info.xt and info.is_float
    out = dec.reconstruct()
    assert np.array_equal(oracle.half_codes_to_float(out), expected(name)), name
    general = bool(dec.xt_params().general)
    assert ("xt_merge" in api.kernel_name(info, xt=dec.xt_params())) or not general


@pytest.mark.gpu
def test_gpu_general_merge_at_4k(oracle, dec):
    """A 4K 4:2:0 frame with a linear R2 ramp and a Q ramp (what the encoder's `fullrange` step adds): kernels against the oracle."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import xt_craft

    if not oracle.have_reference():
        pytest.skip("needs the reference encoder (build container)")
    from libjpeg_amd import synth
    data = oracle.reference_encode_hdr(synth.synth_hdr(1920, 1080, 99), ["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c", "-r12", "-s", "1x1,2x2,2x2"])
    blob = xt_craft.variants(data)["q_and_r2"]
    codes, is_float, err = oracle.decode_xt_status(blob)
    assert err == 0
    dec.read(blob)
    assert np.array_equal(dec.reconstruct(), codes)


def test_lchk_checksum_warning():
    """interface/jpeg.cpp:222-238: a legacy codestream that does not fit the file's LCHK box decodes, with a warning
    (JPGERR_PHASE_ERROR through LastWarning).  The sum (tools/checksum.hpp:104-126, Fletcher modulo 255) runs over the entropy
    coded bytes of the legacy scans without their restart markers: every file the reference encoder wrote must pass."""
    import glob

    import damage

    d = api.Decoder(None)
    files = sorted(glob.glob(os.path.join(GOLDEN_DIR, "xt_*.jpg"))) + sorted(glob.glob(os.path.join(DIR, "enc_*.jpg")))
    assert len(files) >= 10
    flagged = 0
    for fn in files:
        with open(fn, "rb") as f:
            data = f.read()
        d.read(data)
        assert d.last_warning() == (0, None), fn
        b = bytearray(data)
        b[damage.entropy_start(data) + 40] ^= 0x10  # one bit of the legacy scan
        try:
            d.read(bytes(b))
        except api.MijpegError:
            continue  # the flip broke the decode outright
        code, msg = d.last_warning()
        assert code == -1035 and "checksum" in msg, fn
        flagged += 1
    assert flagged >= 8
    d.read(open(os.path.join(GOLDEN_DIR, "pil_200x120_420_dri8.jpg"), "rb").read())  # no LCHK box: nothing to warn about
    assert d.last_warning() == (0, None)
    d.close()
