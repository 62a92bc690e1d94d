"""JPEG::DisplayRectangle as a SEQUENCE of calls on JPEG XT frames (VERDICT r03 missing item 3).

The residual image of an XT frame has row cursors and upsamplers of its own beside the legacy image's
(control/blockbitmaprequester.cpp:228-232, 356-372; PullRData :1118-1146; the residual half of PushReconstructedData :1197-1222
and of ReconstructUnsampled :1054-1071), the two images share m_bSubsampling, and nothing rewinds either: a request shows what
the calls before it left of BOTH.  Contract (include/mijpeg.h, mijpeg_display_rect): what the reference's command line asks
for -- all three components, upsampling and colour transformation on -- with any order and size of rectangles.  A request that
walks a residual cursor without upsampler behind its last row makes the reference dereference a NULL row (it crashes): the
oracle reports OJ_ERR_UNSUPPORTED at that call, the product MIJPEG_ERR_OBJECT_DOESNT_EXIST.

Layers:
  * oracle: oj_xt_requester_new + oj_requester_display against tests/golden/rect_calls_xt/ -- 30 sequences the REAL reference
    library answered (tests/golden/make_rect_calls_xt.py) -- and against that library live on random sequences (CPU);
  * product, host side: the two request models' cursors against the oracle's after every call (CPU, mijpeg_display_plan);
  * -m gpu: the product's pixels through the C ABI and through class JPEG (tests/cxx/rect_calls.cpp on libmijpeg.so).
"""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT
from libjpeg_amd import api
from test_rect_calls import random_script

RC_DIR = os.path.join(GOLDEN_DIR, "rect_calls_xt")
with open(os.path.join(RC_DIR, "manifest.json")) as _f:
    SEQUENCES = json.load(_f)
STREAMS = sorted({v["stream"] for v in SEQUENCES.values()})
OURS = os.path.join(ROOT, "oracle", "_ref", "rect_calls_ours")


def stream(sname):
    with open(os.path.join(RC_DIR, sname + ".jpg"), "rb") as f:
        return f.read()


def golden(name):
    ent = SEQUENCES[name]
    planes = np.fromfile(os.path.join(RC_DIR, name + ".bin"), ent["dtype"]).reshape(ent["shape"])
    return stream(ent["stream"]), [tuple(r) for r in ent["requests"]], planes


def xt_script(rng, w, h, monotone):
    """random_script with the component range and the flags of the contract"""
    return [r[:4] + (0, 2, 1, 1) + r[8:] for r in random_script(rng, w, h, 3, monotone=monotone)]


# ------------------------------------------------------------------------------------------------ oracle
@pytest.mark.parametrize("name", sorted(SEQUENCES))
def test_oracle_requester_equals_what_the_reference_answered(oracle, name):
    data, requests, planes = golden(name)
    _, rcs, canvas, _ = oracle.run_requests_xt(data, requests)
    assert not any(rcs)
    assert canvas.dtype == planes.dtype and np.array_equal(canvas, planes), name


def test_the_goldens_show_the_state_of_both_images(oracle):
    """Skipping the first two stripes of a 4:4:4 / 4:4:4 frame shows the picture's first rows further down (both cursors);
    with a subsampled legacy image the upsampler works by position and only the unsubsampled planes move: neither the picture
    nor the shifted picture."""
    data, _, planes = golden("x444__skip_two")
    plain, _, err = oracle.decode_xt_status(data)
    plain = np.moveaxis(plain, -1, 0)
    assert np.array_equal(planes[:, 16:40], plain[:, 0:24]) and not np.array_equal(planes[:, 16:40], plain[:, 16:40])
    data, _, planes = golden("x420__skip_two")
    plain = np.moveaxis(oracle.decode_xt_status(data)[0], -1, 0)
    assert not np.array_equal(planes[:, 16:40], plain[:, 0:24]) and not np.array_equal(planes[:, 16:40], plain[:, 16:40])
    for s in STREAMS:  # top-down stripes are the plain picture
        data, _, planes = golden(s + "__stripes")
        assert np.array_equal(planes, np.moveaxis(oracle.decode_xt_status(data)[0], -1, 0).astype(planes.dtype)), s


def test_oracle_requester_against_live_reference(oracle):
    """Random sequences on the golden streams: the restatement's bitmaps == the real library's; where the library dies (a NULL
    residual row) the restatement reports the call (build container only)."""
    if not os.path.exists(oracle.REF_RECT_CALLS):
        pytest.skip("oracle/_ref/rect_calls_ref not built")
    from concurrent.futures import ThreadPoolExecutor

    def one(t):
        rng = np.random.default_rng(71000 + t)
        data = stream(STREAMS[t % len(STREAMS)])
        ent = next(v for v in SEQUENCES.values() if v["stream"] == STREAMS[t % len(STREAMS)])
        _, h, w = ent["shape"]
        req = xt_script(rng, w, h, monotone=t % 3 != 0)
        lines, ref = oracle.run_requests_client(oracle.REF_RECT_CALLS, data, req)
        _, rcs, canvas, _ = oracle.run_requests_xt(data, req)
        if ref is None:
            return ("crash", any(rcs))
        return ("ok", not any(rcs) and np.array_equal(ref.astype(canvas.dtype), canvas))

    with ThreadPoolExecutor(8) as ex:
        results = list(ex.map(one, range(120)))
    assert all(ok for _, ok in results), [i for i, (_, ok) in enumerate(results) if not ok]
    assert sum(k == "ok" for k, _ in results) >= 60 and sum(k == "crash" for k, _ in results) >= 5


# ------------------------------------------------------------------------------------------------ product, host side
def test_request_models_follow_both_images(oracle):
    """mijpeg_display_plan on XT streams: the cursors of the legacy AND the residual image after every call equal the oracle's,
    up to the call the reference does not survive."""
    n = 0
    for t in range(60):
        rng = np.random.default_rng(72000 + t)
        sname = STREAMS[t % len(STREAMS)]
        data = stream(sname)
        _, h, w = next(v for v in SEQUENCES.values() if v["stream"] == sname)["shape"]
        req = xt_script(rng, w, h, monotone=t % 2 == 0)
        cur = []
        _, rcs, _, _ = oracle.run_requests_xt(data, req, cursors=cur)
        d = api.Decoder(None)
        d.read_header(data)
        for i, (x0, y0, x1, y1, c0, c1, ups, ct, hm) in enumerate(req):
            if rcs[i]:
                break
            x1 = w - 1 if x1 < 0 else x1
            y1 = h - 1 if y1 < 0 else y1
            plan = d.display_plan(x0, y0, x1, y1, c0, c1, 0, (y0 + hm) if hm else h)
            mine = [plan["comps"][c]["cursor"] for c in range(3)] + [d.display_cursor(4 + c) for c in range(3)]
            assert mine == cur[i], (t, sname, i, req[i])
            n += 1
        d.close()
    assert n >= 150


def test_request_models_recognise_the_plain_picture():
    d = api.Decoder(None)
    d.read_header(stream("x420"))
    assert all(d.display_plan(0, y, 58, min(y + 7, 44), 0, 2, 0, y + 8)["plain"] == 1 for y in range(0, 45, 8))
    d.close()
    d = api.Decoder(None)
    d.read_header(stream("x420"))
    assert d.display_plan(0, 16, 58, 23, 0, 2, 0, 45)["plain"] == 0  # the residual (and luma) cursors stand at row 0
    d.close()


# ------------------------------------------------------------------------------------------------ product, pixels
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(SEQUENCES))
def test_gpu_class_jpeg_answers_like_the_reference(oracle, name):
    """tests/cxx/rect_calls.cpp on top of libmijpeg.so: the same source, the same sequences, the reference's bitmaps."""
    if not os.path.exists(OURS):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_ref/rect_calls_ours"], check=True)
    data, requests, planes = golden(name)
    lines, got = oracle.run_requests_client(OURS, data, requests, env=dict(os.environ))
    assert got is not None, lines
    assert [ln.split()[2:] for ln in lines[1:]] == SEQUENCES[name]["calls"]
    assert np.array_equal(got.astype(planes.dtype), planes), name


@pytest.mark.gpu
def test_gpu_c_abi_random_sequences(oracle):
    """Random sequences through mijpeg_display_rect on every golden stream equal the oracle's restatement (pinned above against
    the real library); the call the reference does not survive fails with OBJECT_DOESNT_EXIST and changes nothing."""
    dec = api.Decoder(0)
    n = refused = 0
    for t in range(100):
        rng = np.random.default_rng(73000 + t)
        sname = STREAMS[t % len(STREAMS)]
        data = stream(sname)
        _, h, w = next(v for v in SEQUENCES.values() if v["stream"] == sname)["shape"]
        req = xt_script(rng, w, h, monotone=t % 3 != 0)
        _, rcs, exp, _ = oracle.run_requests_xt(data, req)
        if any(rcs):  # (the restatement stops in the middle of the call the reference dies in: what stood before it is the expectation)
            exp = oracle.run_requests_xt(data, req[:next(i for i, rc in enumerate(rcs) if rc)])[2]
        dec.read(data, entropy="host" if t % 2 else "auto")
        canvas = np.full(exp.shape, 0xAAAA if exp.dtype == np.uint16 else 0xAA, exp.dtype)
        for i, (x0, y0, x1, y1, c0, c1, ups, ct, hm) in enumerate(req):
            x1 = w - 1 if x1 < 0 else x1
            y1 = h - 1 if y1 < 0 else y1
            if rcs[i]:
                with pytest.raises(api.MijpegError) as e:
                    dec.display_rect(canvas, x0, y0, x1, y1, c0, c1, 0, bm_height=(y0 + hm) if hm else h)
                assert e.value.code == -1031
                refused += 1
                break
            dec.display_rect(canvas, x0, y0, x1, y1, c0, c1, 0, bm_height=(y0 + hm) if hm else h)
        assert np.array_equal(canvas, exp), (t, sname, req)
        n += 1
    dec.close()
    assert n >= 100 and refused >= 3
