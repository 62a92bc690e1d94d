"""JPEG XT (profile C) with ONE component: a grey scale picture with a residual codestream -- `jpeg -r -q .. -Q .. in.pgm`, or an
HDR PFM with one channel.  The reference decodes these with identity transformations throughout (codestream/tables.cpp:2003-2005,
2055-2060, 2079-2081; YCbCrTrafo<.., 1, .., Identity, Identity>, colortrafo/colortransformerfactory.cpp:681-757); round 3 refused
them with MALFORMED_STREAM.  tests/golden/xt_grey/: seven streams with the reference decoder's output
(tests/golden/make_xt_grey.py).  CPU: the oracle against them, the product's host side against the oracle (both codestreams'
coefficients); -m gpu: pixels through the C ABI (xt_merge1_kernel) and the command line."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT
from libjpeg_amd import api

DIR = os.path.join(GOLDEN_DIR, "xt_grey")
with open(os.path.join(DIR, "manifest.json")) as _f:
    CASES = json.load(_f)


def stream(name):
    with open(os.path.join(DIR, name + ".jpg"), "rb") as f:
        return f.read()


def expected(name):
    ent = CASES[name]
    return np.fromfile(os.path.join(DIR, name + ".bin"), ent["dtype"]).reshape(ent["height"], ent["width"])


def as_reference_output(oracle, codes, is_float, dtype):
    codes = codes.reshape(codes.shape[0], codes.shape[1])
    return oracle.half_codes_to_float(codes) if is_float else codes.astype(dtype)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_against_the_reference_decoder(oracle, name):
    codes, is_float, err = oracle.decode_xt_status(stream(name))
    assert err == 0 and codes.shape[2] == 1 and is_float == name.startswith("ghdr")
    want = expected(name)
    assert np.array_equal(as_reference_output(oracle, codes, is_float, want.dtype), want), name


@pytest.mark.parametrize("name", sorted(CASES))
def test_host_side(oracle, name):
    """One component, identity transformations, the output depth of the OCON box; no device needed."""
    d = api.Decoder(None)
    info = d.read(stream(name))
    x = d.xt_params()
    assert info.components == 1 and info.xt == 1 and x.ltrafo_ycbcr == 0 and x.rtrafo_ycbcr == 0 and x.residual.components == 1
    assert info.sample_bytes == (1 if name.startswith("g8") else 2) and bool(info.is_float) == name.startswith("ghdr")
    assert api.kernel_name(info, xt=x) == "idct_planes_kernel+xt_merge1_kernel"
    d.close()


@pytest.mark.parametrize("box,payload", [(b"LTRF", b"\x10"), (b"LTRF", b"\x20"), (b"CTRF", b"\x10")])
def test_a_transformation_box_with_one_component_is_malformed(oracle, box, payload):
    """codestream/tables.cpp:2003-2005, 2079-2081: `Base / Color transformation box exists even though the number of components
    is one` -- the reference binary's verdict where it is here, the oracle's and the product's."""
    import xt_craft

    blob = xt_craft.edit_spec(stream("g8"), xt_craft.subbox(box, payload))
    if oracle.have_reference():
        assert oracle.reference_decode_status(blob)[1] == -1038
    assert oracle.decode_xt_status(blob)[2] == -1038
    d = api.Decoder(None)
    with pytest.raises(api.MijpegError) as e:
        d.read(blob)
    assert e.value.code == -1038
    d.close()


@pytest.fixture(scope="module")
def dec():
    d = api.Decoder(0)
    yield d
    d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("entropy", ["host", "auto"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_pixels_equal_the_reference(oracle, dec, name, entropy):
    info = dec.read(stream(name), entropy=entropy)
    out = dec.reconstruct()
    want = expected(name)
    assert np.array_equal(as_reference_output(oracle, out, bool(info.is_float), want.dtype), want), name


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["g8", "g16", "ghdr"])
def test_gpu_cli_writes_the_references_file(oracle, tmp_path, name):
    src = tmp_path / "in.jpg"
    src.write_bytes(stream(name))
    dst = tmp_path / ("out.pfm" if name.startswith("ghdr") else "out.pgm")
    cli = os.path.join(ROOT, "libjpeg_amd", "bin", "jpeg")
    subprocess.run([cli, str(src), str(dst)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    got = oracle.read_pfm_reference(str(dst)).astype("<f4") if name.startswith("ghdr") else oracle.read_pnm_any(str(dst))
    assert np.array_equal(got.reshape(expected(name).shape), expected(name)), name


@pytest.mark.gpu
def test_gpu_grey_xt_at_full_hd(oracle, dec):
    """1920 x 1080 against the oracle (pinned above on the small ones)."""
    if not oracle.have_reference():
        pytest.skip("needs the reference encoder (build container)")
    import tempfile

    from libjpeg_amd import synth

    g = synth.synth_image(1920, 1080, 3, channels=1).reshape(1080, 1920)
    with tempfile.TemporaryDirectory() as d:
        src, dst = os.path.join(d, "in.pgm"), os.path.join(d, "out.jpg")
        with open(src, "wb") as f:
            f.write(b"P5\n1920 1080\n255\n" + g.tobytes())
        subprocess.run([oracle.REF_BIN, "-r", "-q", "85", "-Q", "90", src, dst], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        data = open(dst, "rb").read()
    codes, is_float, err = oracle.decode_xt_status(data)
    assert err == 0 and not is_float
    dec.read(data)
    out = dec.reconstruct()
    assert np.array_equal(out.reshape(1080, 1920), codes.reshape(1080, 1920).astype(out.dtype))
