"""JPEG XT decoded WITHOUT colour transformation: `jpeg -c in.jpg out`, JPGTAG_MATRIX_LTRAFO = NONE of a DisplayRectangle request,
MIJPEG_FLAG_NO_COLOR_TRANSFORM.  The reference builds its transformer with `disabletorgb`
(control/blockbitmaprequester.cpp:1251, colortrafo/colortransformerfactory.cpp:231-232): the STANDARD YCbCr L transformation
becomes the identity -- a free-form matrix stays -- and the rest of the merge (L tables, residual chain, C transformation, output
conversion) is unchanged.  tests/golden/xt_noct/: nine streams with the reference decoder's `-c` output
(tests/golden/make_xt_noct.py).  CPU: the oracle against those and the host's `ltrafo_standard`; -m gpu: the product's pixels
through the C ABI and the command line."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT
from libjpeg_amd import api

DIR = os.path.join(GOLDEN_DIR, "xt_noct")
with open(os.path.join(DIR, "manifest.json")) as _f:
    CASES = json.load(_f)


def stream(name):
    with open(os.path.join(DIR, name + ".jpg"), "rb") as f:
        return f.read()


def expected(name):
    ent = CASES[name]
    return np.fromfile(os.path.join(DIR, name + ".bin"), ent["dtype"]).reshape(ent["height"], ent["width"], 3)


def as_reference_output(oracle, codes, is_float):
    return oracle.half_codes_to_float(codes) if is_float else codes.astype(np.uint8)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_against_the_reference_decoder(oracle, name):
    codes, is_float, err = oracle.decode_xt_status(stream(name), no_color_transform=True)
    assert err == 0
    assert np.array_equal(as_reference_output(oracle, codes, is_float), expected(name)), name
    plain = oracle.decode_xt_status(stream(name))[0]
    assert (not np.array_equal(codes, plain)) == CASES[name]["differs_from_the_plain_decode"]  # (the free-form matrix: no change)


@pytest.mark.parametrize("name", sorted(CASES))
def test_host_side_knows_which_l_transformation_the_switch_replaces(name):
    d = api.Decoder(None)
    d.read(stream(name))
    x = d.xt_params()
    assert x.ltrafo_ycbcr == 1 and x.ltrafo_standard == (0 if name == "hxyz" else 1)
    d.close()


@pytest.fixture(scope="module")
def dec():
    d = api.Decoder(0)
    yield d
    d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_pixels_equal_the_reference(oracle, dec, name):
    info = dec.read(stream(name))
    out = dec.reconstruct(api.FLAG_NO_COLOR_TRANSFORM)
    assert np.array_equal(as_reference_output(oracle, out, bool(info.is_float)), expected(name)), name
    # the fused kernels carry the YCbCr transformation: they step aside for the identity
    name_c = api.kernel_name(info, api.FLAG_NO_COLOR_TRANSFORM, xt=dec.xt_params())
    assert ("fusedxt" in name_c) == (name == "hxyz" and "fusedxt" in api.kernel_name(info, xt=dec.xt_params())), name_c
    # and the plain decode is still the plain decode (the cached frame is per flag set)
    plain = oracle.decode_xt_status(stream(name))[0]
    assert np.array_equal(dec.reconstruct(), plain.astype(out.dtype)), name


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["h420", "i420"])
def test_gpu_cli_with_c_writes_the_references_file(oracle, tmp_path, name):
    """`jpeg -c in.jpg out` of libjpeg_amd/bin/jpeg: the reference's PFM / PPM samples"""
    src = tmp_path / "in.jpg"
    src.write_bytes(stream(name))
    dst = tmp_path / ("out.pfm" if name[0] == "h" else "out.ppm")
    cli = os.path.join(ROOT, "libjpeg_amd", "bin", "jpeg")
    subprocess.run([cli, "-c", str(src), str(dst)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    got = oracle.read_pfm_reference(str(dst)).astype("<f4") if name[0] == "h" else oracle.read_pnm_any(str(dst))
    assert np.array_equal(got, expected(name)), name
