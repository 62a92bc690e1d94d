"""JPEG::DisplayRectangle as a SEQUENCE of calls.

The reference keeps state between the calls of one JPEG object: a row cursor per component that never rewinds and the
line buffers of its upsamplers (control/blockbitmaprequester.cpp:1013-1272, upsampling/upsamplerbase.cpp:138-327).  What a
request shows therefore depends on the requests before it -- most visibly in the reference's own command line, which
writes frames of two or four components component by component (cmd/reconstruct.cpp:272-303) and gets planes of zeros for
the later unsubsampled components of a frame that also has a subsampled one.  Layers:

  * oracle: oj_requester (a literal restatement with real line buffers) against tests/golden/rect_calls/ -- sequences the
    REAL reference library answered (tests/golden/make_rect_calls.py) -- and, where oracle/_ref/rect_calls_ref exists, against
    that binary live on random layouts and random sequences (CPU);
  * product, host side: request_model.hpp's cursors against the oracle's after every call (CPU, mijpeg_display_plan);
  * -m gpu: the product's pixels -- tests/cxx/rect_calls.cpp linked with libmijpeg.so, the C ABI through ctypes, the
    `jpeg` front end and the reference's own client on top of libmijpeg.so -- against the reference's (oracle/_ref/* travel
    to the GPU box) on >= 200 random layouts of one to four components.
"""
import json
import os
import subprocess

import numpy as np
import pytest

import craft
from conftest import GOLDEN_DIR, ROOT
from libjpeg_amd import api

RC_DIR = os.path.join(GOLDEN_DIR, "rect_calls")
with open(os.path.join(RC_DIR, "manifest.json")) as _f:
    SEQUENCES = json.load(_f)
OURS = os.path.join(ROOT, "oracle", "_ref", "rect_calls_ours")


def golden(name):
    ent = SEQUENCES[name]
    with open(os.path.join(RC_DIR, name + ".jpg"), "rb") as f:
        data = f.read()
    planes = np.fromfile(os.path.join(RC_DIR, name + ".bin"), ent["dtype"]).reshape(ent["shape"])
    return data, [tuple(r) for r in ent["requests"]], planes


def random_script(rng, w, h, nc, fullwidth=True, aligned=False, monotone=False):
    """Request sequences no sane client issues: any order of stripes, heights, component ranges, bitmap heights."""
    req, y = [], 0
    for _ in range(int(rng.integers(1, 12))):
        if monotone:
            if y >= h:
                break
            y0 = y
        else:
            y0 = int(rng.integers(0, h))
        if aligned:
            y0 &= ~7
        y1 = min(h - 1, y0 + int(rng.choice([1, 3, 8, 8, 8, 16, 24, 40])) - 1)
        y = y1 + 1
        c0 = int(rng.integers(0, nc))
        c1 = int(rng.integers(c0, nc))
        if rng.random() < 0.4:
            c0, c1 = 0, nc - 1
        ups = 1 if rng.random() < 0.8 else 0
        if not ups:
            c1 = c0
        x0, x1 = (0, -1) if fullwidth else (lambda a: (a, int(rng.integers(a, w))))(int(rng.integers(0, w)))
        req.append((x0, y0, x1, y1, c0, c1, ups, int(rng.integers(0, 2)), int(rng.choice([0, 0, 8, 16, 5]))))
    return req


def cli_scripts(w, h, samp):
    """The reference command line's own sequences: PPM stripes, PGX with and without upsampling (cmd/reconstruct.cpp:272-342)."""
    nc = len(samp)
    vmax = max(s[1] for s in samp)
    return {
        "stripes": [(0, y, -1, min(y + 7, h - 1), 0, nc - 1, 1, 1, 8) for y in range(0, h, 8)],
        "pgx": [(0, y, -1, min(y + 7, h - 1), c, c, 1, 1, 8) for c in range(nc) for y in range(0, h, 8)],
        "pgx_noup": [(0, y, -1, min(y + 8 * (vmax // samp[c][1]) - 1, h - 1), c, c, 0, 0, 0) for c in range(nc)
                     for y in range(0, h, 8 * (vmax // samp[c][1]))],
        "whole": [(0, 0, -1, -1, 0, nc - 1, 1, 1, 0)],
    }


# ------------------------------------------------------------------------------------------------ oracle
@pytest.mark.parametrize("name", sorted(SEQUENCES))
def test_oracle_requester_equals_what_the_reference_answered(oracle, name):
    data, requests, planes = golden(name)
    _, rcs, canvas = oracle.run_requests(data, requests)
    assert not any(rcs)
    assert np.array_equal(canvas, planes), name


def test_the_goldens_show_the_state_between_calls(oracle):
    """The fixtures are not the plain picture: the PGX loops leave planes of zeros, the backwards order leaves other ones."""
    zero_planes = 0
    for name in SEQUENCES:
        if not name.endswith("_pgx"):
            continue
        data, _, planes = golden(name)
        plain = np.moveaxis(oracle.decode(data), -1, 0)
        for c in range(planes.shape[0]):
            if not planes[c].any() and plain[c].any():
                zero_planes += 1
    assert zero_planes >= 6
    data, _, fwd = golden("c4_22_11_11_22_pgx")
    _, _, bwd = golden("c4_22_11_11_22_pgx_backwards")
    assert not np.array_equal(fwd, bwd)


def test_oracle_requester_against_live_reference(oracle):
    """Random layouts (1..4 components, factors 1..4 x 1..4, sizes 1..140, DRI), the command line's sequences and random
    ones: the restatement's bitmaps == the real library's (build container only)."""
    if not os.path.exists(oracle.REF_RECT_CALLS):
        pytest.skip("oracle/_ref/rect_calls_ref not built")
    from concurrent.futures import ThreadPoolExecutor

    def one(t):
        rng = np.random.default_rng(41000 + t)
        samp, w, h, dri = craft.random_layout(rng)
        data = craft.craft_stream(rng, samp, w, h, dri)
        scripts = dict(cli_scripts(w, h, samp)) if t % 3 == 0 else {}
        scripts["random"] = random_script(rng, w, h, len(samp), aligned=t % 4 == 1, monotone=t % 2 == 0)
        bad = []
        for sname, req in scripts.items():
            lines, ref = oracle.run_requests_client(oracle.REF_RECT_CALLS, data, req)
            _, rcs, ora = oracle.run_requests(data, req)
            if ref is None or any(rcs) or not np.array_equal(ref, ora):
                bad.append((t, samp, w, h, sname))
        return bad

    with ThreadPoolExecutor(8) as ex:
        bad = [b for r in ex.map(one, range(240)) for b in r]
    assert not bad, bad[:5]


# ------------------------------------------------------------------------------------------------ product, host side
def test_request_model_cursors_follow_the_oracle():
    """mijpeg_display_plan advances the product's request state without a device: after every call of 400 random sequences
    (partial widths, no upsampling, bitmaps lower than the stripe) every component's cursor stands where the oracle's does."""
    from oracle import oracle as O
    for t in range(400):
        rng = np.random.default_rng(7000 + t)
        samp, w, h, dri = craft.random_layout(rng)
        data = craft.craft_stream(rng, samp, w, h, dri)
        req = random_script(rng, w, h, len(samp), fullwidth=False, aligned=(t % 3 == 0), monotone=(t % 2 == 0))
        cur = []
        O.run_requests(data, req, cursors=cur)
        d = api.Decoder(None)
        d.read_header(data)
        for i, (x0, y0, x1, y1, c0, c1, ups, ct, hm) in enumerate(req):
            x1 = w - 1 if x1 < 0 else x1
            y1 = h - 1 if y1 < 0 else y1
            plan = d.display_plan(x0, y0, x1, y1, c0, c1, (0 if ct else api.FLAG_NO_COLOR_TRANSFORM) | (0 if ups else api.FLAG_NO_UPSAMPLING),
                                  (y0 + hm) if hm else h)
            assert [plan["comps"][c]["cursor"] for c in range(len(samp))] == cur[i], (t, samp, w, h, i, req[i])
        d.close()


def test_request_model_recognises_the_plain_picture():
    """Top-down stripes over all components are the plain picture (served from the cached frame); the second pass of the PGX
    loop over an unsubsampled component of a mixed frame is not: its rows are zeros."""
    rng = np.random.default_rng(1)
    data = craft.craft_stream(rng, [(1, 1), (2, 1)], 50, 40)
    d = api.Decoder(None)
    d.read_header(data)
    for y in range(0, 40, 8):
        assert d.display_plan(0, y, 49, y + 7, 0, 0, 0, y + 8)["plain"] == 1  # component 0 (subsampled): the picture
    plans = [d.display_plan(0, y, 49, y + 7, 1, 1, 0, y + 8) for y in range(0, 40, 8)]
    assert all(p["plain"] == 0 and p["comps"][1]["zeros"] == 1 for p in plans)
    d.close()
    d = api.Decoder(None)
    d.read_header(data)
    assert all(d.display_plan(0, y, 49, y + 7, 0, 1, 0, y + 8)["plain"] == 1 for y in range(0, 40, 8))
    d.close()


# ------------------------------------------------------------------------------------------------ product, pixels
def _client_env():
    return dict(os.environ)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(SEQUENCES))
def test_gpu_class_jpeg_answers_like_the_reference(oracle, name):
    """tests/cxx/rect_calls.cpp on top of libmijpeg.so: the same source, the same sequences, the reference's bitmaps."""
    if not os.path.exists(OURS):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/rect_calls_ours"], check=True)
    data, requests, planes = golden(name)
    lines, got = oracle.run_requests_client(OURS, data, requests, env=_client_env())
    assert got is not None, lines
    assert [ln.split()[2:] for ln in lines[1:]] == SEQUENCES[name]["calls"]
    assert np.array_equal(got, planes), name


@pytest.mark.gpu
def test_gpu_c_abi_sequences_on_random_layouts(oracle):
    """>= 200 random layouts of one to four components through mijpeg_display_rect: the command line's sequences and random
    ones (unaligned corners, repeated and skipped stripes, component subsets, low bitmaps) equal the oracle's restatement,
    which the CPU tests pin against the real library."""
    dec = api.Decoder(0)
    n = 0
    for t in range(220):
        rng = np.random.default_rng(52000 + t)
        samp, w, h, dri = craft.random_layout(rng)
        data = craft.craft_stream(rng, samp, w, h, dri)
        scripts = dict(cli_scripts(w, h, samp)) if t % 2 == 0 else {}
        scripts["random"] = random_script(rng, w, h, len(samp), aligned=t % 4 == 1, monotone=t % 3 != 0)
        for sname, req in scripts.items():
            _, rcs, exp = oracle.run_requests(data, req)
            assert not any(rcs)
            f = dec.read(data, entropy="host")
            canvas = np.full(exp.shape, 0xAA, exp.dtype)
            for (x0, y0, x1, y1, c0, c1, ups, ct, hm) in req:
                x1 = w - 1 if x1 < 0 else x1
                y1 = h - 1 if y1 < 0 else y1
                flags = (0 if ct else api.FLAG_NO_COLOR_TRANSFORM) | (0 if ups else api.FLAG_NO_UPSAMPLING)
                try:
                    dec.display_rect(canvas, x0, y0, x1, y1, c0, c1, flags, bm_height=(y0 + hm) if hm else h)
                except api.MijpegError as e:  # the class refuses what the reference refuses (component ranges without upsampling)
                    assert e.code == -1024 and not ups and c0 != c1
            assert np.array_equal(canvas, exp), (t, samp, w, h, dri, sname)
            n += 1
    dec.close()
    assert n >= 500


@pytest.mark.gpu
def test_gpu_command_lines_write_the_references_planes(oracle, tmp_path):
    """libjpeg_amd/bin/jpeg AND the reference's own cmd/reconstruct.cpp on top of libmijpeg.so (oracle/_ref/jpeg_dropin) against
    the reference binary's output files on >= 200 random layouts of one to four components (PPM for one and three, PGX planes
    otherwise): byte-equal.  Where oracle/_ref/jpeg is absent the oracle's command-line restatement stands in."""
    ours = os.path.join(ROOT, "libjpeg_amd", "bin", "jpeg")
    dropin = os.path.join(ROOT, "oracle", "_ref", "jpeg_dropin")
    clients = [ours] + ([dropin] if os.path.exists(dropin) else [])
    use_ref = oracle.have_reference()
    n = mixed = 0
    for t in range(210):
        rng = np.random.default_rng(63000 + t)
        samp, w, h, dri = craft.random_layout(rng)
        data = craft.craft_stream(rng, samp, w, h, dri)
        if use_ref:
            exp, err = oracle.reference_decode_status(data)
        else:
            exp, err, _ = oracle.decode_status(data)
        assert err == 0, (t, samp, err)
        src = tmp_path / "in.jpg"
        src.write_bytes(data)
        for exe in clients:
            dst = tmp_path / "out.ppm"
            r = subprocess.run([exe, str(src), str(dst)], capture_output=True, timeout=120)
            assert r.returncode == 0, (exe, r.stderr[-300:])
            got = _read_cli_output(str(dst), oracle)
            assert got.shape == exp.shape and np.array_equal(got, exp), (t, samp, w, h, dri, os.path.basename(exe))
        n += 1
        nc = len(samp)
        mixed += int(nc in (2, 4) and len({s for s in samp}) > 1)
    assert n >= 200 and mixed >= 30


def _read_cli_output(path, oracle):
    import re
    with open(path, "rb") as f:
        head = f.read(2)
    if head in (b"P5", b"P6"):
        return oracle.read_pnm_any(path)
    planes = []
    for line in open(path).read().split():
        m = re.match(rb"PG ML \+(\d+) (\d+) (\d+)", open(line[:-4] + ".h", "rb").read())
        bits, w, h = (int(x) for x in m.groups())
        planes.append(np.fromfile(line, np.uint8 if bits <= 8 else ">u2").reshape(h, w))
    return np.stack(planes, axis=-1)
