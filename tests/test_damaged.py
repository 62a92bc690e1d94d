"""Damaged codestreams: same input, same result as the reference.

The reference does not give up on a damaged entropy-coded segment: EntropyParser::ParseRestartMarker
(codestream/entropyparser.cpp:117-201) resynchronises at the restart markers (a marker that is BEHIND the expected one is
dropped and the search goes on, one that is AHEAD invalidates the segment, whose MCUs are zeroed,
codestream/sequentialscan.cpp:416-420), the bit reader feeds zero bits at markers and at the end of the data
(io/bitstream.cpp:56-137), refinement scans warn and continue (codestream/refinementscan.cpp:667).  Three layers:

  * the oracle's restatement of that behaviour against committed goldens (tests/golden/damaged/, written by the real
    reference through tests/golden/make_damaged.py) and, where oracle/_ref/jpeg exists, against the binary itself on
    seeded corruptions (CPU);
  * the product's host entropy decoder against the oracle on seeded corruptions, coefficient by coefficient (CPU);
  * -m gpu: the product end to end (host decoder + reconstruction kernels, through the C ABI) against the reference
    binary (it travels to the GPU box as oracle/_ref/jpeg; the oracle stands in where it is absent): error code AND pixels.
"""
import json
import os
import struct

import numpy as np
import pytest

import damage
from conftest import GOLDEN_DIR, MANIFEST, golden_jpeg
from libjpeg_amd import api

DAMAGED_DIR = os.path.join(GOLDEN_DIR, "damaged")
with open(os.path.join(DAMAGED_DIR, "manifest.json")) as _f:
    DAMAGED = json.load(_f)

# plain JPEG fixtures, the DNL ones among them (damaged JPEG XT streams: tests/test_xt_damaged.py)
BASES = sorted(k for k, v in MANIFEST.items() if not v.get("big") and v.get("kind") != "xt_float32")


def damaged_jpeg(name):
    with open(os.path.join(DAMAGED_DIR, name + ".jpg"), "rb") as f:
        return f.read()


def damaged_pixels(name):
    ent = DAMAGED[name]
    if ent["error"] != 0:
        return None
    with open(os.path.join(DAMAGED_DIR, name + ".bin"), "rb") as f:
        return np.frombuffer(f.read(), np.uint8).reshape(ent["height"], ent["width"], ent["channels"])


def seeded(count_per_file, seed, where="any", bases=BASES):
    for fi, name in enumerate(bases):
        data = golden_jpeg(name)
        for kind, blob in damage.cases(data, count_per_file, seed * 1000 + fi, where):
            yield name, kind, blob


# ------------------------------------------------------------------------------------------------ oracle
@pytest.mark.parametrize("name", sorted(DAMAGED))
def test_oracle_on_handmade_damage(oracle, name):
    """The restatement against what the real reference wrote for the committed hand-made cases."""
    ent = DAMAGED[name]
    px, err, _ = oracle.decode_status(damaged_jpeg(name))
    assert err == ent["error"], f"{name}: reference {ent['error']}, oracle {err}"
    if err == 0:
        assert np.array_equal(px, damaged_pixels(name)), name


def test_handmade_damage_differs_from_the_clean_picture(oracle):
    """The fixtures are not trivially equal to the undamaged decode (the resynchronisation really is exercised)."""
    differing = 0
    for name, ent in DAMAGED.items():
        if ent["error"] == 0:
            clean = oracle.decode(golden_jpeg(ent["base"]))
            differing += int(not np.array_equal(clean, damaged_pixels(name)))
    assert differing >= 8


def test_oracle_against_live_reference_on_seeded_damage(oracle):
    """600 seeded corruptions: verdict and pixels of the restatement == the real binary's (build container only)."""
    if not oracle.have_reference():
        pytest.skip("oracle/_ref/jpeg not built")
    from concurrent.futures import ThreadPoolExecutor

    work = list(seeded(15, 42))[:600]

    def one(item):
        name, kind, blob = item
        opx, oerr, _ = oracle.decode_status(blob)
        if oerr is None:
            return None
        rpx, rerr = oracle.reference_decode_status(blob)
        if rerr == "timeout":
            return None
        if rerr != oerr:
            return f"{name}/{kind}: reference {rerr}, oracle {oerr}"
        if rerr == 0 and not (rpx.shape == opx.shape and np.array_equal(rpx, opx)):
            return f"{name}/{kind}: pixels differ"
        return ""

    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(one, work))
    bad = [r for r in res if r]
    assert not bad, bad[:10]
    assert sum(r == "" for r in res) >= 500


# ------------------------------------------------------------------------------------------------ product, host side
@pytest.mark.parametrize("name", sorted(DAMAGED))
def test_host_decoder_on_handmade_damage(oracle, name):
    verdict, detail = damage.product_vs_oracle(damaged_jpeg(name))
    assert verdict == "ok", (name, verdict, detail)
    # and the error code is the reference's own
    ent = DAMAGED[name]
    d = api.Decoder(None)
    try:
        try:
            d.read(damaged_jpeg(name))
            code = 0
        except api.MijpegError as e:
            code = e.code
    finally:
        d.close()
    assert code == ent["error"]


def test_host_decoder_against_oracle_on_seeded_damage(oracle):
    """Coefficients, quantiser tables and error codes of the product's host decoder on 1200 seeded corruptions."""
    stats = {}
    bad = []
    for where, seed in (("any", 5), ("entropy", 6)):
        for name, kind, blob in seeded(18, seed, where):
            verdict, detail = damage.product_vs_oracle(blob)
            stats[verdict] = stats.get(verdict, 0) + 1
            if verdict not in ("ok", "skip"):
                bad.append((name, kind, verdict, detail))
    assert not bad, bad[:10]
    assert stats.get("ok", 0) >= 1100, stats


def test_parallel_and_sequential_host_walks_agree_on_damage(oracle):
    """Thread count must not matter: the restart-parallel plan falls back to the sequential walk on any irregularity."""
    d1, d8 = api.Decoder(None), api.Decoder(None)
    n = 0
    for name, kind, blob in seeded(6, 9, "entropy", ["pil_200x120_420_dri8", "ref_75x45_420_dri2", "ref_97x61_3x3", "refprog_64x64_444_dri5"]):
        res = []
        for d, t in ((d1, 1), (d8, 8)):
            try:
                f = d.read(blob, threads=t)
                res.append((0, [d.coefficients(c) for c in range(f.components)]))
            except api.MijpegError as e:
                res.append((e.code, []))
        assert res[0][0] == res[1][0], (name, kind)
        for a, b in zip(res[0][1], res[1][1]):
            assert np.array_equal(a, b), (name, kind)
        n += 1
    d1.close()
    d8.close()
    assert n == 24


# ------------------------------------------------------------------------------------------------ product, end to end
@pytest.fixture(scope="module")
def dec():
    d = api.Decoder(0)
    yield d
    d.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(DAMAGED))
def test_gpu_handmade_damage_pixels_and_rc(dec, name):
    ent = DAMAGED[name]
    verdict, detail = damage.product_pixels_vs_expected(dec, damaged_jpeg(name), damaged_pixels(name), ent["error"])
    assert verdict == "ok", (name, verdict, detail)


@pytest.mark.gpu
def test_gpu_seeded_damage_pixels_and_rc_equal_the_reference(oracle, dec):
    """>= 500 seeded corruptions (byte flips, dropped / duplicated / reordered intervals, renumbered and removed markers,
    truncation, insertions, zero and FF runs) of the plain JPEG fixtures: the product's return code and pixels equal those of
    oracle/_ref/jpeg, the real reference binary.  Where the binary is absent the oracle's restatement (pinned against it by
    the CPU tests above and tools/damage_campaign.py) stands in."""
    from concurrent.futures import ThreadPoolExecutor

    use_ref = oracle.have_reference()
    work = list(seeded(10, 77)) + list(seeded(10, 78, "entropy"))
    with ThreadPoolExecutor(16) as ex:
        expected = list(ex.map(lambda w: damage.expected_of(w[2], use_ref), work))
    stats = {}
    bad = []
    for (name, kind, blob), (epx, eerr) in zip(work, expected):
        verdict, detail = damage.product_pixels_vs_expected(dec, blob, epx, eerr)
        stats[verdict] = stats.get(verdict, 0) + 1
        if verdict not in ("ok", "skip"):
            bad.append((name, kind, verdict, detail))
    print("damaged-stream parity against", "oracle/_ref/jpeg" if use_ref else "the oracle", stats)
    assert not bad, bad[:10]
    assert stats.get("ok", 0) >= 500, stats


@pytest.mark.gpu
def test_gpu_device_entropy_decoder_never_differs_on_damage(oracle, dec):
    """entropy="prefer-gpu": the on-device Huffman decoder may only ever answer what the host walk (= the reference) answers.
    Damaged intervals -- among them ones whose codes stay valid but that run past their data (the bit-addressed reader
    supplies zero bits there) -- must be reported and handed to the host: return code and pixels equal the reference's."""
    from libjpeg_amd import synth
    streams = [golden_jpeg(n) for n in ("pil_200x120_420_dri8", "ref_75x45_420_dri2", "pil_33x17_420_dri1")]
    streams.append(synth.synth_jpeg(320, 200, 3, 85, "420", 2))
    use_ref = oracle.have_reference()
    stats, on_device = {}, 0
    for si, data in enumerate(streams):
        rng = np.random.default_rng(900 + si)
        for kind in ("truncate", "zero_run", "ff_run", "flip1", "flipbits", "insert", "del1", "drop_interval", "ff00_to_ffxx", "rst_burst") * 8:
            blob = damage.corrupt(data, kind, rng, "entropy")
            epx, eerr = damage.expected_of(blob, use_ref)
            if eerr is None:
                continue
            try:
                dec.read(blob, entropy="prefer-gpu")
                perr = 0
                on_device += dec.entropy_used == "gpu"
            except api.MijpegError as e:
                perr = e.code
            assert (perr == 0) == (eerr == 0) and (perr == eerr or perr == 0), (si, kind, eerr, perr)
            if perr == 0:
                assert np.array_equal(dec.reconstruct_cli(), epx), (si, kind, dec.entropy_used)
            stats[kind] = stats.get(kind, 0) + 1
    assert sum(stats.values()) >= 250, stats
    assert on_device >= 20  # some damage leaves a stream the device decodes like the reference does; most goes to the host


@pytest.mark.gpu
def test_gpu_damaged_big_frame_through_the_fused_kernel(oracle, dec):
    """A frame large enough for the restart-parallel host plan and the fused 4:2:0 kernel, with intervals dropped,
    duplicated and a marker renumbered: pixels equal the oracle's (and the reference's, where present)."""
    from libjpeg_amd import synth
    data = synth.synth_jpeg(640, 368, 31, 85, "420", 4)
    rng = np.random.default_rng(5)
    for kind in ("drop_interval", "dup_interval", "swap_intervals", "renumber_rst", "drop_marker_only", "flip3", "truncate"):
        blob = damage.corrupt(data, kind, rng, "entropy")
        epx, eerr = damage.expected_of(blob, oracle.have_reference())
        opx, oerr, _ = oracle.decode_status(blob)
        assert oerr == eerr and (eerr != 0 or np.array_equal(opx, epx)), kind
        verdict, detail = damage.product_pixels_vs_expected(dec, blob, epx, eerr)
        assert verdict == "ok", (kind, verdict, detail)


# ------------------------------------------------------------------------------------------------ coefficients beyond 16 bits
def _runaway_cases():
    """Frames whose DC predictions wander far beyond +-32767 (every code legal, damage.runaway_dc): 4:2:0, 4:4:4, 4:2:2, grey."""
    import io

    from PIL import Image

    from libjpeg_amd import synth
    out = []
    for i, (sub, w, h) in enumerate((("420", 200, 120), ("444", 136, 72), ("422", 168, 96), ("gray", 160, 88), ("420", 648, 368))):
        if sub == "gray":
            b = io.BytesIO()
            Image.fromarray(synth.synth_image(w, h, 40 + i)).convert("L").save(b, "JPEG", quality=85)
            base = b.getvalue()
        else:
            base = synth.synth_jpeg(w, h, 40 + i, 85, sub, 0)
        out.append((f"{sub}_{w}x{h}", damage.runaway_dc(base, 7 + i)))
    return out


def test_runaway_dc_keeps_32_bit_coefficients_like_the_reference(oracle):
    """The reference stores LONG coefficients: a DC prediction beyond 16 bits is not an error.  The host decoder turns to
    int32 planes for such a frame (info.coef_wide, mijpeg_coefficients32); oracle, reference binary and product agree."""
    for name, blob in _runaway_cases():
        opx, oerr, _ = oracle.decode_status(blob)
        assert oerr == 0, name
        if oracle.have_reference():
            rpx, rerr = oracle.reference_decode_status(blob)
            assert rerr == 0 and np.array_equal(rpx.reshape(opx.shape), opx), name
        d = api.Decoder(None)
        try:
            f = d.read(blob, entropy="host")
            assert f.coef_wide == 1 and f.fast_arith == 0, name
            assert max(int(np.abs(d.coefficients(c)).max()) for c in range(f.components)) > 32767, name
            assert max(f.range_max[c] for c in range(f.components)) > 32767, name
        finally:
            d.close()
        verdict, detail = damage.product_vs_oracle(blob)
        assert verdict == "ok", (name, verdict, detail)


@pytest.mark.gpu
def test_gpu_runaway_dc_pixels_equal_the_reference(oracle, dec):
    """... and the unfused kernels reconstruct the int32 planes with the reference's 32-bit transform (idct_planes_long_kernel):
    pixels equal the reference's; a sound frame decoded by the same object afterwards is back on the 16-bit store and the
    fused kernel; the device entropy decoder declines such a stream (MIJPEG_ERR_NOT_AVAILABLE) without touching anything."""
    from libjpeg_amd import synth
    for name, blob in _runaway_cases():
        epx, eerr = damage.expected_of(blob, oracle.have_reference())
        assert eerr == 0, name
        verdict, detail = damage.product_pixels_vs_expected(dec, blob, epx.reshape(oracle.decode(blob).shape), eerr)
        assert verdict == "ok", (name, verdict, detail)
        assert dec.info.coef_wide == 1 and api.kernel_name(dec.info) == "idct_planes_long_kernel+upsample_color_kernel", name
        with pytest.raises(api.MijpegError) as e:
            dec.read(blob, entropy="gpu")
        assert e.value.code == -1029, name
        f = dec.read(blob, entropy="prefer-gpu")  # the host decoder takes over
        assert f.coef_wide == 1
        assert np.array_equal(dec.reconstruct(), epx.reshape(oracle.decode(blob).shape)), name
    good = synth.synth_jpeg(200, 120, 3, 85, "420", 0)
    f = dec.read(good, entropy="host")
    assert f.coef_wide == 0
    assert np.array_equal(dec.reconstruct(), oracle.decode(good))


# ------------------------------------------------------------------------------------------------ JPEG LS markers in a DCT file
def _segment(marker, payload):
    return bytes([0xff, marker]) + struct.pack(">H", len(payload) + 2) + payload


def _ls_marker_cases(data):
    """LSE (0xfff8) and the long DRI flavours of JPEG LS around a DCT frame: name -> (stream, the reference's verdict; None =
    outside the accelerated subset).  The header part of a file is read before its frame type is known
    (codestream/decoder.cpp:85 passes isls = true): the markers are legal there and nowhere else."""
    i = 2
    while data[i + 1] != 0xda:
        i += 2 + ((data[i + 2] << 8) | data[i + 3])
    sos = i
    three = data[[k for k in range(2, sos) if data[k] == 0xff and data[k + 1] in (0xc0, 0xc1, 0xc2)][0] + 9] == 3
    put = lambda at, s: data[:at] + s + data[at:]  # noqa: E731
    thresholds = _segment(0xf8, b"\x01" + struct.pack(">5H", 255, 3, 7, 21, 64))
    trafo = lambda depth, flags=0: _segment(0xf8, b"\x0d" + struct.pack(">H", 255) + bytes([depth]) + bytes(range(1, depth + 1)) +  # noqa: E731
                                            (bytes([flags]) + b"\x00\x00" * (depth - 1)) * depth)
    own = [k for k in range(2, sos) if data[k] == 0xff and data[k + 1] == 0xdd and data[k + 2:k + 4] == b"\x00\x04"]
    long_dri = bytes([0xff, 0xdd, 0, 6, 0, 0]) + (data[own[-1] + 4:own[-1] + 6] if own else b"\x00\x00")  # (the file's own interval)
    return {
        "thresholds": (put(2, thresholds), 0),
        "unknown_id": (put(2, _segment(0xf8, b"\x07abcdef")), 0),
        "length_2": (put(2, _segment(0xf8, b"")), -1038),
        "length_3": (put(2, _segment(0xf8, b"\x07")), 0),
        "length_4": (put(2, _segment(0xf8, b"\x07\x00")), 0),
        "thresholds_short": (put(2, _segment(0xf8, b"\x01" + struct.pack(">4H", 255, 3, 7, 21))), -1038),
        "mapping_table": (put(2, _segment(0xf8, b"\x02\x01\x01\x00\x00")), -1034),
        "size_extension": (put(2, _segment(0xf8, b"\x04\x04\x00\x00\x00\x10")), -1034),
        "trafo_for_one": (put(2, trafo(1)), None if three else 0),
        "trafo_for_three": (put(2, trafo(3)), None if three else 0),
        "trafo_twice": (put(2, trafo(1) + trafo(1)), -1038),
        "trafo_shift_33": (put(2, trafo(1, 0x21)), -1028),
        "trafo_of_nothing": (put(2, _segment(0xf8, b"\x0d" + struct.pack(">H", 255) + b"\x00")), -1038),
        "behind_the_frame_header": (put(sos, thresholds), -1038),
        "dri_6_bytes": (put(2, long_dri), 0),
        "dri_5_bytes": (put(2, long_dri[:3] + b"\x05" + long_dri[5:]), 0),
        "dri_6_bytes_beyond_16_bits": (put(2, long_dri[:5] + b"\x01" + long_dri[6:]), None),
        "dri_6_bytes_behind_the_frame_header": (put(sos, long_dri), -1038),
        "dri_6_bytes_behind_one_in_the_header_part": (put(2, bytes([0xff, 0xdd, 0, 4, 0, 9]))[:sos + 6] + long_dri + data[sos:], 0),
    }


@pytest.mark.parametrize("base", ["ref_80x48_420", "pil_70x40_gray", "refprog_97x61_420"])
def test_jpeg_ls_markers_in_the_header_part(oracle, base):
    clean = oracle.decode(golden_jpeg(base))
    for name, (blob, want) in _ls_marker_cases(golden_jpeg(base)).items():
        px, oerr, _ = oracle.decode_status(blob)
        d = api.Decoder(None)
        try:
            d.read(blob, entropy="host")
            perr = 0
        except api.MijpegError as e:
            perr = e.code
        finally:
            d.close()
        if want is None:  # declined by both: the LS colour transformation on a DCT frame, 32 bit restart intervals
            assert perr == -1034 and oerr in (None, -1034), (name, oerr, perr)
            continue
        assert (oerr, perr) == (want, want), (name, oerr, perr)
        if want == 0:
            assert np.array_equal(px, clean), name
            verdict, detail = damage.product_vs_oracle(blob)
            assert verdict == "ok", (name, verdict, detail)
        if oracle.have_reference():
            rpx, rerr = oracle.reference_decode_status(blob)
            assert rerr == want, (name, rerr)
            if want == 0:
                assert np.array_equal(rpx, px), name


@pytest.mark.parametrize("base", ["ref_80x48_420", "refprog_97x61_420"])
def test_differential_frame_types_are_malformed_not_declined(oracle, base):
    """SOF5-7 / SOF13-15 without a DHP marker in front: "found a differential frame outside a hierarchical image process"
    (codestream/image.cpp:487-500) before the header is read -- the reference's verdict, not a coding process to decline."""
    data = golden_jpeg(base)
    at = [k for k in range(2, len(data)) if data[k] == 0xff and data[k + 1] in (0xc0, 0xc1, 0xc2)][0]
    for marker in (0xc5, 0xc6, 0xc7, 0xcd, 0xce, 0xcf):
        blob = data[:at + 1] + bytes([marker]) + data[at + 2:]
        assert oracle.decode_status(blob)[1] == -1038, hex(marker)
        d = api.Decoder(None)
        with pytest.raises(api.MijpegError) as e:
            d.read(blob)
        d.close()
        assert e.value.code == -1038, hex(marker)
        if oracle.have_reference():
            assert oracle.reference_decode_status(blob)[1] == -1038, hex(marker)
