"""JPEG XT profile C with an INTEGER output of eight bits: what the reference encoder writes
for `jpeg -r -q .. -Q .. in.ppm` -- 8-bit legacy codestream, residual codestream, an output conversion (OCON box) without extra
range bits.  The merge is the one of the 16-bit files with every constant derived from the output depth
(colortrafo/colortransformerfactory.cpp:300-372: m_lOutMax = 255, the Q / R2 / L tables scaled to 8 + 4 resp. 8 bits,
colortrafo/ycbcrtrafo.cpp:700-850 with clamping instead of the half-float cast).

tests/golden/xt_int8/ holds the streams and the REAL reference decoder's PPM samples (or error code) for each
(tests/golden/make_xt_int8.py).  CPU: the oracle against those goldens; the product's host side against the oracle (error codes,
output depth, table sizes).  -m gpu: the product's pixels through the C-ABI, host and device entropy decode."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from libjpeg_amd import api

DIR = os.path.join(GOLDEN_DIR, "xt_int8")
with open(os.path.join(DIR, "manifest.json")) as _f:
    CASES = json.load(_f)
GOOD = sorted(k for k, v in CASES.items() if v["error"] == 0)


def stream(name):
    with open(os.path.join(DIR, name + ".jpg"), "rb") as f:
        return f.read()


def expected(name):
    ent = CASES[name]
    with open(os.path.join(DIR, name + ".bin"), "rb") as f:
        return np.frombuffer(f.read(), np.uint8).reshape(ent["height"], ent["width"], 3)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_against_the_reference_decoder(oracle, name):
    codes, is_float, err = oracle.decode_xt_status(stream(name))
    assert err == CASES[name]["error"], (name, err)
    if err == 0:
        assert not is_float and int(codes.max()) <= 255
        assert np.array_equal(codes.astype(np.uint8), expected(name)), name


@pytest.mark.parametrize("name", sorted(CASES))
def test_host_side_output_depth(name):
    """Error codes are the reference's; what decodes asks the kernels for one byte per sample and carries real Q / R2 tables
    (the arithmetic identities of the kernels are those of a 16-bit output)."""
    d = api.Decoder(None)
    try:
        try:
            info = d.read(stream(name))
            code = 0
        except api.MijpegError as e:
            code = e.code
        assert code == CASES[name]["error"], name
        if code == 0:
            x = d.xt_params()
            assert info.xt and not info.is_float and info.sample_bytes == 1
            assert x.out_max == 255 and x.out_shift == 128 and x.general == 1
            rprec = x.residual.precision + x.residual_hidden_bits
            assert x.qtable_entries == 1 << (rprec + 4)
            for c in range(3):
                assert x.qtable[c] and x.r2table[c]
                r2 = np.ctypeslib.as_array((C.c_int32 * 4096).from_address(x.r2table[c]))
                if "r2_" not in name and "q_and_r2" not in name:
                    assert np.array_equal(r2, (np.arange(4096) + 8) >> 4), name  # the default identity, 12 bits -> 8
    finally:
        d.close()


@pytest.fixture(scope="module")
def dec():
    d = api.Decoder(0)
    yield d
    d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("entropy", ["host", "auto"])
@pytest.mark.parametrize("name", GOOD)
def test_gpu_pixels_equal_the_reference(dec, name, entropy):
    info = dec.read(stream(name), entropy=entropy)
    assert info.xt and not info.is_float and info.sample_bytes == 1
    out = dec.reconstruct()
    assert out.dtype == np.uint8
    assert np.array_equal(out, expected(name)), name
    assert "xt_merge_general" in api.kernel_name(info, xt=dec.xt_params())


@pytest.mark.gpu
def test_gpu_cli_writes_the_references_ppm(tmp_path):
    """bin/jpeg (the reference's cmd/reconstruct.cpp loop over class JPEG) writes a P6 file with maxval 255, byte for byte the
    reference's."""
    import subprocess

    src, dst = tmp_path / "in.jpg", tmp_path / "out.ppm"
    src.write_bytes(stream("enc_420"))
    cli = os.path.join(os.path.dirname(api.__file__), "bin", "jpeg")
    subprocess.run([cli, str(src), str(dst)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    blob = dst.read_bytes()
    want = expected("enc_420")
    head = b"P6\n%d %d\n255\n" % (want.shape[1], want.shape[0])
    assert blob.startswith(head) and blob[len(head):] == want.tobytes()


@pytest.mark.gpu
def test_gpu_full_hd_round_trip(oracle, dec):
    """Full size: a 1920 x 1080 4:2:0 file against the oracle (pinned above on the small ones); the reference's lossy residual at
    -Q 90 brings the picture back to within 3 code values of the source."""
    if not oracle.have_reference():
        pytest.skip("needs the reference encoder (build container)")
    from libjpeg_amd import synth

    img = synth.synth_image(1920, 1080, 11)
    data = oracle.reference_encode(img, ["-r", "-q", "85", "-Q", "90", "-s", "1x1,2x2,2x2"])
    codes, is_float, err = oracle.decode_xt_status(data)
    assert err == 0 and not is_float
    dec.read(data)
    out = dec.reconstruct()
    assert out.dtype == np.uint8 and np.array_equal(out, codes.astype(np.uint8))


# ------------------------------------------------------------------------------------------------ 16-bit integer output
# The same files from a PPM with 16-bit samples: 8 extra range bits, clamping, no cast to float (tests/golden/make_xt_int16.py;
# the reference decoder's PPM samples, plain and with -c).  4:2:0 runs the fused XT kernel with its integer clamp.
DIR16 = os.path.join(GOLDEN_DIR, "xt_int16")
with open(os.path.join(DIR16, "manifest.json")) as _f:
    CASES16 = json.load(_f)


def stream16(name):
    with open(os.path.join(DIR16, name + ".jpg"), "rb") as f:
        return f.read()


def expected16(name, tag):
    ent = CASES16[name]
    return np.fromfile(os.path.join(DIR16, f"{name}.{tag}.bin"), "<u2").reshape(ent["height"], ent["width"], 3)


@pytest.mark.parametrize("noct", [False, True])
@pytest.mark.parametrize("name", sorted(CASES16))
def test_oracle_16bit_integer_output(oracle, name, noct):
    codes, is_float, err = oracle.decode_xt_status(stream16(name), no_color_transform=noct)
    assert err == 0 and not is_float
    assert np.array_equal(codes, expected16(name, "noct" if noct else "plain")), name


@pytest.mark.gpu
@pytest.mark.parametrize("noct", [False, True])
@pytest.mark.parametrize("name", sorted(CASES16))
def test_gpu_16bit_integer_output(dec, name, noct):
    info = dec.read(stream16(name))
    assert info.xt and not info.is_float and info.sample_bytes == 2 and dec.xt_params().out_max == 65535
    out = dec.reconstruct(api.FLAG_NO_COLOR_TRANSFORM if noct else 0)
    assert out.dtype == np.uint16 and np.array_equal(out, expected16(name, "noct" if noct else "plain")), name
    if name == "w420_r12" and not noct:
        assert api.kernel_name(info, xt=dec.xt_params()) == "fusedxt420_kernel"
