"""A merging specification WITHOUT a residual codestream that asks for more than the plain picture: what the reference's encoder
writes for `jpeg -q .. -R n -h in` (no `-r`) from an HDR or 16-bit picture -- hidden refinement bits, an L table from 8 + n bits to
the output's depth, an output conversion.  The reference builds the Extended transformer with R transformation "zero"
(colortrafo/colortransformerfactory.cpp:262-283) and shows the legacy picture through the L chain alone (L transformation, L table,
C transformation, clamp / half float; colortrafo/ycbcrtrafo.cpp:744-746, 861-878).  Round 4 refused all of them (-1034).
tests/golden/xt_lonly/: 56 streams with the reference decoder's output (tests/golden/make_xt_lonly.py).
CPU: the oracle against them and against the live binary on a sweep, the product's host side against the oracle (coefficients,
parameters); -m gpu: pixels through the C ABI, the command line's files."""
import json
import os
import subprocess
import tempfile

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT
from libjpeg_amd import api

DIR = os.path.join(GOLDEN_DIR, "xt_lonly")
with open(os.path.join(DIR, "manifest.json")) as _f:
    CASES = json.load(_f)


def stream(name):
    with open(os.path.join(DIR, name + ".jpg"), "rb") as f:
        return f.read()


def expected(name):
    ent = CASES[name]
    return np.fromfile(os.path.join(DIR, name + ".bin"), "<u2").reshape(ent["height"], ent["width"], ent["channels"])


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_against_the_reference_decoder(oracle, name):
    codes, is_float, err = oracle.decode_xt_status(stream(name))
    assert err == 0 and is_float == CASES[name]["is_float"]
    assert np.array_equal(codes, expected(name)), name


@pytest.mark.parametrize("name", sorted(CASES))
def test_host_side(oracle, name):
    """The product's host decoder: an XT frame without residual planes, the legacy coefficients with their hidden bits, the L
    table the oracle builds."""
    ent = CASES[name]
    d = api.Decoder(None)
    info = d.read(stream(name))
    x = d.xt_params()
    assert info.xt == 1 and info.components == ent["channels"] and info.sample_bytes == 2 and bool(info.is_float) == ent["is_float"]
    assert x.no_residual == 1 and x.residual.components == 0 and x.hidden_bits == ent["hidden"] and x.ltable_entries == 256 << ent["hidden"]
    assert x.out_max == 65535 and x.general == 0
    assert api.kernel_name(info, xt=x) == ("idct_planes_kernel+xt_merge1_kernel" if ent["channels"] == 1 else "idct_planes_kernel+xt_merge_kernel")
    oinfo, planes = oracle.decode_xt_planes(stream(name))
    assert oinfo.precision == 8 + ent["hidden"]
    for c in range(info.components):
        assert np.array_equal(d.coefficients(c).astype(np.int32), planes[c]), (name, c)
    d.close()


def _sweep_sources(rng, w, h):
    from libjpeg_amd import synth

    hdr = synth.synth_hdr(w, h, int(rng.integers(1, 1 << 30))).astype("<f4") * float(rng.choice([0.25, 1.0, 8.0]))
    i16 = (np.clip(hdr / hdr.max(), 0, 1) ** 0.45 * 65535).astype(np.uint16)
    return [
        (b"PF\n%d %d\n-1.0\n" % (w, h), hdr[::-1].tobytes(), "pfm", 3),
        (b"Pf\n%d %d\n-1.0\n" % (w, h), hdr[:, :, 0][::-1].tobytes(), "pfm", 1),
        (b"P6\n%d %d\n65535\n" % (w, h), i16.astype(">u2").tobytes(), "ppm", 3),
        (b"P5\n%d %d\n65535\n" % (w, h), i16[:, :, 2].astype(">u2").tobytes(), "pgm", 1),
    ]


def _live_cases(oracle, count, seed):
    """Reference-encoded files (random size, quality, hidden bits, sampling, scan layout) with the reference decoder's output."""
    rng = np.random.default_rng(seed)
    tmp = "/dev/shm" if os.path.isdir("/dev/shm") else None
    made = 0
    while made < count:
        w, h = int(rng.integers(1, 70)), int(rng.integers(1, 60))
        header, raw, ext, ch = _sweep_sources(rng, w, h)[int(rng.integers(0, 4))]
        args = ["-q", str(int(rng.integers(20, 98))), "-R", str(int(rng.integers(1, 5))), "-h"]
        if rng.integers(0, 3) == 0:
            args += ["-v"]
        if rng.integers(0, 3) == 0:
            args += ["-z", str(int(rng.integers(1, 9)))]
        if ch == 3 and rng.integers(0, 2):
            args += ["-s", str(rng.choice(["1x1,2x2,2x2", "1x1,2x1,2x1", "1x1,1x2,1x2", "1x1,3x1,3x1", "1x1,4x2,4x2"]))]
        if rng.integers(0, 4) == 0:
            args += ["-c"]  # no colour transformation at the encoder: SPEC{LTRF = identity}
        dec_args = ["-c"] if ch == 3 and rng.integers(0, 5) == 0 else []
        with tempfile.TemporaryDirectory(dir=tmp) as d:
            src, jpg, dst = os.path.join(d, "in." + ext), os.path.join(d, "x.jpg"), os.path.join(d, "out." + ext)
            with open(src, "wb") as f:
                f.write(header + raw)
            r = subprocess.run([oracle.REF_BIN, *args, src, jpg], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            if r.returncode or b"failed" in r.stderr:
                continue
            blob = open(jpg, "rb").read()
            r = subprocess.run([oracle.REF_BIN, *dec_args, jpg, dst], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            assert r.returncode == 0 and b"failed" not in r.stderr, (args, r.stderr)
            if ext == "pfm":
                want = oracle.read_pfm_reference(dst).astype("<f2").view("<u2").reshape(h, w, ch)
            else:
                want = oracle.read_pnm_any(dst).reshape(h, w, ch)
        made += 1
        yield blob, want, bool(dec_args), (w, h, args, dec_args)


def test_oracle_against_live_reference(oracle):
    if not oracle.have_reference():
        pytest.skip("oracle/_ref/jpeg not built")
    n = 0
    for blob, want, noct, what in _live_cases(oracle, 120, 20260926):
        codes, is_float, err = oracle.decode_xt_status(blob, no_color_transform=noct)
        assert err == 0, what
        assert np.array_equal(codes, want), what
        n += 1
    assert n == 120


def test_specification_errors_without_a_residual(oracle):
    """What the transformer refuses without a residual frame: R transformation zero / JPEG_LS (Tables::RTrafoTypeOf,
    codestream/tables.cpp:2040-2075), output without clamping (no such transformer: colortransformerfactory.cpp:698-725, 850-885),
    an L table that does not exist; the reference binary's verdict where it is here, the oracle's and the product's."""
    import xt_craft

    base = stream("i16_R2_seq")  # (16-bit integer output: the binary writes a PPM the helper can read)
    cases = [
        (xt_craft.edit_spec(base, xt_craft.subbox(b"RTRF", b"\x00")), -1038),
        (xt_craft.edit_spec(base, xt_craft.subbox(b"RTRF", b"\x30")), -1038),
        (xt_craft.edit_spec(base, xt_craft.subbox(b"RTRF", b"\x10")), 0),  # never looked at without a residual
    ]
    for blob, want in cases:
        if oracle.have_reference():
            assert oracle.reference_decode_status(blob)[1] == want
        assert oracle.decode_xt_status(blob)[2] == want
        d = api.Decoder(None)
        try:
            d.read(blob)
            code = 0
        except api.MijpegError as e:
            code = e.code
        d.close()
        assert code == want


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
    assert out.dtype == np.uint16 and np.array_equal(out.reshape(expected(name).shape), expected(name)), name


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["hdr_R2_seq", "hdr_R4_420", "ghdr_R3_prog", "i16_R1_z2", "i16_R4_420", "g16_R2_seq"])
def test_gpu_cli_writes_the_references_file(oracle, tmp_path, name):
    ent = CASES[name]
    src = tmp_path / "in.jpg"
    src.write_bytes(stream(name))
    ext = "pfm" if ent["is_float"] else ("ppm" if ent["channels"] == 3 else "pgm")
    dst = tmp_path / ("out." + ext)
    cli = os.path.join(ROOT, "libjpeg_amd", "bin", "jpeg")
    subprocess.run([cli, str(src), str(dst)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if ent["is_float"]:
        got = oracle.read_pfm_reference(str(dst))
        assert got.shape[2] == ent["channels"]
        assert np.array_equal(got.astype("<f2").view("<u2"), expected(name)) and np.array_equal(got.astype("<f2").astype(np.float32), got)
    else:
        assert np.array_equal(oracle.read_pnm_any(str(dst)).reshape(expected(name).shape), expected(name))


@pytest.mark.gpu
def test_gpu_no_colour_transformation(oracle, dec):
    """`jpeg -c`: the standard YCbCr L transformation becomes the identity, the L tables stay (colortransformerfactory.cpp:231-232)."""
    for name in ("hdr_R2_seq", "i16_R3_420"):
        codes, _, err = oracle.decode_xt_status(stream(name), no_color_transform=True)
        assert err == 0
        dec.read(stream(name))
        out = dec.reconstruct(flags=api.FLAG_NO_COLOR_TRANSFORM)
        assert np.array_equal(out.reshape(codes.shape), codes), name


@pytest.mark.gpu
def test_gpu_live_sweep_against_the_oracle(oracle, dec):
    """Random reference-encoded files (where the encoder is here) through the kernels: all samplings, -c at either end."""
    if not oracle.have_reference():
        pytest.skip("needs the reference encoder (build container)")
    for blob, want, noct, what in _live_cases(oracle, 40, 7):
        dec.read(blob)
        out = dec.reconstruct(flags=api.FLAG_NO_COLOR_TRANSFORM if noct else 0)
        assert np.array_equal(out.reshape(want.shape), want), what


@pytest.mark.gpu
def test_gpu_lonly_at_4k(oracle, dec):
    """3840 x 2160 HDR with two hidden bits, 4:2:0, against the oracle (pinned above on the small ones)."""
    if not oracle.have_reference():
        pytest.skip("needs the reference encoder (build container)")
    from libjpeg_amd import synth

    hdr = synth.synth_hdr(3840, 2160, 99)
    data = oracle.reference_encode_hdr(hdr, ["-q", "85", "-R", "2", "-h", "-s", "1x1,2x2,2x2"])
    codes, is_float, err = oracle.decode_xt_status(data)
    assert err == 0 and is_float
    dec.read(data)
    out = dec.reconstruct()
    assert np.array_equal(out, codes)


def test_an_eob_run_above_hidden_bits_is_out_of_sync(oracle):
    """The visible scans of a sequential frame with hidden bits sit above them (Al + hidden), but the reference's parser for them is
    not a progressive one (m_bProgressive needs lowbit > hidden, sequentialscan.cpp:84-87): an EOB-run symbol there -- a DHT value
    turned into 0x40 -- is "AC coefficient decoding out of sync" (-1038, :722-750), not a run of empty blocks (tools/box_campaign.py
    r5: the scan pipeline of the host decoder took it for one)."""
    data = bytearray(stream("i16_R1_z2"))
    dht = data.rindex(b"\xff\xc4")
    i = data.index(b"\x82", dht + 40)
    data[i] = 0x40
    data = bytes(data)
    if oracle.have_reference():
        assert oracle.reference_decode_status(data)[1] == -1038
    assert oracle.decode_xt_status(data)[2] == -1038
    d = api.Decoder(None)
    with pytest.raises(api.MijpegError) as e:
        d.read(data)
    assert e.value.code == -1038
    d.close()


def test_a_tone_box_that_never_completes_is_an_empty_table_zero(oracle):
    """A TONE box whose announced length never fills up stays in the reference's list as an object nobody parsed: no entries and a
    table index nobody initialised (boxes/tonemapperbox.hpp:64-76) -- in practice the zero of fresh heap memory.  A specification
    that names table 0 finds it and its ScaledTableOf refuses: INVALID_PARAMETER, not OBJECT_DOESNT_EXIST
    (inversetonemappingbox.cpp:192-212; the most frequent mismatch of tools/box_campaign.py r5 until it was restated)."""
    for name in ("i16_R3_prog", "hdr_R2_seq", "g16_R1_seq"):
        data = bytearray(stream(name))
        i = data.index(b"TONE") - 4  # LBox in front of TBox
        data[i] ^= 0x40
        data = bytes(data)
        if oracle.have_reference():
            assert oracle.reference_decode_status(data)[1] == -1024, name
        assert oracle.decode_xt_status(data)[2] == -1024, name
        d = api.Decoder(None)
        with pytest.raises(api.MijpegError) as e:
            d.read(data)
        assert e.value.code == -1024, name
        d.close()


def test_errors_come_in_stream_order(oracle):
    """A DHT value turned into a symbol that runs the scan into the next marker (-1025 from the bit reader) in front of a scan header
    whose length is wrong (-1038): the reference reads in order and reports the first; the oracle's header walk over all scans (it
    collects the boxes) used to report the second."""
    data = bytearray(stream("hdr_R1_prog"))
    sos = [i for i in range(len(data) - 1) if data[i] == 0xFF and data[i + 1] == 0xDA]
    a, b = 2034, 2146  # two of the visible scans' headers
    assert a in sos and b in sos and data[a - 1] == 0x31
    data[a - 1] = 0xCC  # the last value of the DHT segment in front of scan a
    one = bytes(data)
    data[b + 2] = 0x83  # ... and the length field of scan header b
    both = bytes(data)
    for blob in (one, both):
        if oracle.have_reference():
            assert oracle.reference_decode_status(blob)[1] == -1025
        assert oracle.decode_xt_status(blob)[2] == -1025
        d = api.Decoder(None)
        with pytest.raises(api.MijpegError) as e:
            d.read(blob)
        assert e.value.code == -1025
        d.close()


def _three_way(oracle, blob, want):
    if oracle.have_reference():
        assert oracle.reference_decode_status(blob)[1] == want
    assert oracle.decode_xt_status(blob)[2] == want
    d = api.Decoder(None)
    with pytest.raises(api.MijpegError) as e:
        d.read(blob)
    assert e.value.code == want
    d.close()


def test_a_box_segment_the_file_ends_in_is_filled_with_zeros(oracle):
    """DecoderStream::Append (io/decoderstream.cpp:136-158) gives a box every byte its segment announced: what the file no longer
    had is zeros.  A merging specification cut short is therefore "found a box size of zero within a superbox" (-1038), not an EOF
    inside a box header (-1025, what rounds 4-5 answered; tools/box_campaign.py r5)."""
    data = stream("g16_R1_seq")
    i = data.index(b"SPEC")
    _three_way(oracle, data[:i + 4 + 15], -1038)


def test_hidden_scans_give_their_verdict_before_the_transformer_refuses(oracle):
    """What the colour transformer refuses arrives with the first request; the reference has read the whole file by then, the hidden
    refinement scans of the FINE boxes included (marker/frame.cpp:1063-1070).  A TONE box that never completes (-1024 at the request)
    AND a damaged refinement scan: the scan's -1038 comes first."""
    data = bytearray(stream("i16_R3_prog"))
    t = data.index(b"TONE")
    seg = t - 16  # FF EB of the TONE box's segment
    assert data[seg:seg + 2] == b"\xff\xeb"
    data[seg + 3] -= 5  # the segment is five bytes shorter than the box needs: the box never completes
    only_tone = bytes(data)
    _three_way(oracle, only_tone, -1024)
    data[5145] = 0xC3  # inside the refinement scan of the FINE box with En = 5
    _three_way(oracle, bytes(data), -1038)
