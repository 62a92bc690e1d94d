"""Pins the CPU restatement (oracle/jpeg_oracle.c) against the REAL reference.

1. every committed fixture in tests/golden (pixels produced by the reference binary, see
   tests/golden/make_golden.py) must be reproduced bit-exactly;
2. when oracle/_ref/jpeg is available (build container), fresh random streams are decoded by both
   and compared live.
These run on CPU (-m "not gpu")."""
import hashlib

import numpy as np
import pytest

from conftest import BIG_CASES, MANIFEST, P12_CASES, SMALL_CASES, XT_CASES, big_jpeg, golden_jpeg, golden_pixels
from libjpeg_amd import synth


@pytest.mark.parametrize("name", SMALL_CASES)
def test_oracle_matches_reference_golden(oracle, name):
    ent = MANIFEST[name]
    out = oracle.decode(golden_jpeg(name))
    assert out.shape == (ent["height"], ent["width"], ent["channels"])
    exp = golden_pixels(name)
    if exp is not None:
        assert np.array_equal(out, exp), f"{(out != exp).sum()} differing samples"
    assert hashlib.sha256(out.tobytes()).hexdigest() == ent["pixels_sha256"]


@pytest.mark.parametrize("name", [n for n in BIG_CASES if "4k" in n])
def test_oracle_matches_reference_4k(oracle, name):
    data = big_jpeg(name)
    if data is None:
        pytest.skip("this box's Pillow produces different bytes than the manifest recipe")
    out = oracle.decode(data)
    assert hashlib.sha256(out.tobytes()).hexdigest() == MANIFEST[name]["pixels_sha256"]


@pytest.mark.parametrize("name", XT_CASES)
def test_oracle_xt_profile_c_matches_reference_golden(oracle, name):
    """JPEG XT profile C: half-float codes, expanded exactly like cmd/bitmaphook.cpp:282-305 does, must equal the floats
    the reference wrote into its PFM, bit for bit."""
    ent = MANIFEST[name]
    codes, is_float = oracle.decode_xt(golden_jpeg(name))
    assert is_float and codes.shape == (ent["height"], ent["width"], 3)
    out = oracle.half_codes_to_float(codes)
    exp = golden_pixels(name)
    if exp is not None:
        assert np.array_equal(out.view(np.uint32), exp.view(np.uint32))
    assert hashlib.sha256(np.ascontiguousarray(out, "<f4").tobytes()).hexdigest() == ent["pixels_sha256"]


@pytest.mark.parametrize("name", P12_CASES)
def test_oracle_12bit_matches_reference_golden(oracle, name):
    ent = MANIFEST[name]
    out = oracle.decode16(golden_jpeg(name))
    exp = golden_pixels(name)
    if exp is not None:
        assert np.array_equal(out, exp)
    assert hashlib.sha256(np.ascontiguousarray(out, "<u2").tobytes()).hexdigest() == ent["pixels_sha256"]


def test_oracle_xt_vs_live_reference(oracle):
    if not oracle.have_reference():
        pytest.skip("oracle/_ref/jpeg not built (needs /root/reference)")
    for w, h, extra in [(97, 61, []), (160, 112, ["-s", "1x1,2x2,2x2"]), (48, 80, ["-q", "40", "-Q", "50"])]:
        data = oracle.reference_encode_hdr(synth.synth_hdr(w, h, w) * 3.0, ["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c", "-r12"] + extra)
        codes, _ = oracle.decode_xt(data)
        ref = oracle.reference_decode_hdr(data)
        assert np.array_equal(oracle.half_codes_to_float(codes).view(np.uint32), ref.view(np.uint32))


def test_scan_order_and_quant_natural_order(oracle):
    # DQT values are stored de-zigzagged (marker/quantization.cpp:502-527): the first row of the
    # natural-order table must be zig-zag positions 0,1,5,6,14,15,27,28
    data = golden_jpeg("pil_80x48_444")
    i = data.index(b"\xff\xdb")
    zz = np.frombuffer(data[i + 5:i + 5 + 64], np.uint8)
    info = oracle.read_info(data)
    q0 = np.array(info.quant[0][:])
    assert list(q0[:8]) == [int(zz[k]) for k in (0, 1, 5, 6, 14, 15, 27, 28)]


def test_idct_dc_only_and_null(oracle):
    # DC-only block: every sample equals ((dc*q*16 + 128*128) ... ) through both passes; check
    # against the closed form of dct/idct.cpp with all AC = 0
    q = np.full(64, 8, np.uint16)
    c = np.zeros(64, np.int32)
    c[0] = 5
    out = oracle.idct_block(c, q)
    s0 = 5 * (8 << 4) + (128 << 7)
    row = ((s0 << 9) + 256) >> 9
    val = ((row << 9) + 2048) >> 12
    assert (out == val).all()


@pytest.mark.parametrize("sx,sy", [(2, 2), (2, 1), (1, 2)])
def test_upsample_inplace_alias_quirk(oracle, sx, sy):
    # out[1] of every 8-wide group is computed from the already overwritten out[2]
    # (upsampling/upsampler.cpp:301-302); a closed form WITHOUT the alias must differ somewhere.
    rng = np.random.default_rng(0)
    plane = rng.integers(0, 4096, size=(16, 16)).astype(np.int32)
    blk = oracle.upsample_block(plane, 16, 16, sx, sy, 8, 8)
    if sx == 2:
        y = 8 // sy
        x0 = 8 // 2
        def V(j, l):
            cc = np.clip(x0 - 1 + j, 0, 15)
            if sy == 1:
                return int(plane[min(y + l, 15), cc])
            yy = y + l // 2
            if l % 2 == 0:
                return (int(plane[max(yy - 1, 0), cc]) + 3 * int(plane[yy, cc]) + (2 if j % 2 == 0 else 1)) >> 2
            return (int(plane[min(yy + 1, 15), cc]) + 3 * int(plane[yy, cc]) + (1 if j % 2 == 0 else 2)) >> 2
        for l in range(8):
            s = [V(j + 1, l) for j in range(-1, 6)]  # s[k] = src[k-1+1] ... index shift: s[0]=src[-1]
            src = lambda k: s[k + 1]
            o2 = (src(0) + 3 * src(1) + 2) >> 2
            assert blk[l, 2] == o2
            assert blk[l, 1] == (o2 + 3 * src(0) + 1) >> 2
            assert blk[l, 0] == (src(-1) + 3 * src(0) + 2) >> 2
            assert blk[l, 7] == (src(4) + 3 * src(3) + 1) >> 2


def _live_cases():
    rng = np.random.default_rng(99)
    for i in range(12):
        w, h = int(rng.integers(1, 200)), int(rng.integers(1, 150))
        yield w, h, int(rng.integers(30, 98)), ["444", "420", "422"][i % 3], int(rng.integers(0, 5)), 1000 + i


@pytest.mark.parametrize("w,h,q,sub,dri,seed", list(_live_cases()))
def test_oracle_vs_live_reference(oracle, w, h, q, sub, dri, seed):
    if not oracle.have_reference():
        pytest.skip("oracle/_ref/jpeg not built (needs /root/reference)")
    data = synth.encode_jpeg(synth.synth_image(w, h, seed), q, sub, dri)
    assert np.array_equal(oracle.decode(data), oracle.reference_decode(data))


@pytest.mark.parametrize("w,h,sub,dri,scale", [(200, 120, "420", 8, 16), (97, 61, "444", 0, 16), (129, 71, "422", 3, 9), (64, 48, "420", 0, 40), (33, 200, "420", 1, 3)])
def test_oracle_vs_live_reference_12bit(oracle, w, h, sub, dri, scale):
    """12-bit streams made by synth.to_12bit (SOF1, P = 12, 16-bit quantiser entries over an 8-bit stream's entropy coded
    data): the restatement against the reference binary -- these streams are the 12-bit content of the GPU parity tests and of
    tools/layout_bench.py.  With `scale` 16 the picture is the 8-bit one times sixteen, give or take the rounding."""
    j8 = synth.synth_jpeg(w, h, 300 + w, 85, sub, dri)
    data = synth.to_12bit(j8, scale)
    px = oracle.decode16(data)
    assert px.dtype == np.uint16 and px.shape == (h, w, 3) and px.max() <= 4095
    if scale == 16:
        assert np.abs(px.astype(int) - 16 * oracle.decode(j8).astype(int)).max() <= 24
    if not oracle.have_reference():
        pytest.skip("oracle/_ref/jpeg not built (needs /root/reference)")
    rpx, rerr = oracle.reference_decode_status(data)
    assert rerr == 0 and np.array_equal(np.asarray(rpx).reshape(px.shape), px)


# ------------------------------------------------------------------------------------------------------
# encoder direction: forward colour transformation + box downsampling + forward DCT + quantiser
# ------------------------------------------------------------------------------------------------------
def _covered(info, planes, c):
    """The blocks that cover samples (the MCU padding blocks belong to the entropy coder)."""
    return planes[c][:(info.ch[c] + 7) // 8, :(info.cw[c] + 7) // 8]


FORWARD_GOLDEN = [n for n, e in MANIFEST.items() if e.get("encoder") == "ref" and e.get("channels") == 3 and "seed" in e
                  and "-n" not in e["args"] and "-c" not in e["args"] and "sof1" not in n and not n.startswith("xt_") and "lumasub" not in n]
# (-c: no forward colour transformation; oracle.forward(info, img, 1) applies one)


@pytest.mark.parametrize("name", FORWARD_GOLDEN)
def test_forward_matches_the_reference_encoders_golden_files(oracle, name):
    """The committed golden files were written by the reference ENCODER from synth_image(w, h, seed): the coefficients in
    them pin the forward restatement (colour transformation, downsampler edges, FDCT, the single-precision quantiser)."""
    ent = MANIFEST[name]
    img = synth.synth_image(ent["width"], ent["height"], ent["seed"])
    info, ref = oracle.decode_coefficients(golden_jpeg(name))
    mine = oracle.forward(info, img, 1)
    for c in range(info.ncomp):
        assert np.array_equal(_covered(info, mine, c), _covered(info, ref, c)), (name, c)


@pytest.mark.parametrize("w,h,args,seed", [(512, 512, ["-bl", "-q", "75"], 1234), (640, 360, ["-bl", "-q", "97", "-s", "1x1,2x2,2x2"], 5),
                                           (333, 211, ["-bl", "-q", "40", "-s", "1x1,2x1,2x1"], 6), (70, 40, ["-bl", "-q", "90"], 7)])
def test_forward_matches_the_live_reference_encoder(oracle, w, h, args, seed):
    if not oracle.have_reference():
        pytest.skip("needs oracle/_ref/jpeg")
    rng = np.random.default_rng(seed)
    img = synth.synth_image(w, h, seed) if seed != 6 else rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    if seed == 7:
        img = np.ascontiguousarray(img[..., :1])
    info, ref = oracle.decode_coefficients(oracle.reference_encode(img.squeeze(), args))
    mine = oracle.forward(info, img, 1)
    for c in range(info.ncomp):
        assert np.array_equal(_covered(info, mine, c), _covered(info, ref, c)), c


def test_forward_then_inverse_is_close(oracle):
    """Sanity of the pair: forward at a fine quantiser, then the (pinned) inverse path, reproduces the picture closely."""
    img = synth.synth_image(96, 64, 9)
    info, _ = oracle.decode_coefficients(golden_jpeg("ref_64x40_q100"))
    info.width, info.height = 96, 64
    for c in range(3):
        info.cw[c], info.ch[c], info.bw[c], info.bh[c] = 96, 64, 12, 8
    info.mcus_x, info.mcus_y = 12, 8
    planes = oracle.forward(info, img, 1)
    back = oracle.reconstruct(info, planes)
    err = np.abs(back.astype(int) - img.astype(int))
    assert err.max() <= 4 and err.mean() < 0.8, (err.max(), err.mean())

