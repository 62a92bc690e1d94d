"""CPU tests (-m "not gpu") of the host side of the product: the C-ABI library loads without a GPU and
exports every symbol include/mijpeg.h declares; header parsing and the restart-interval-parallel
Huffman decoder reproduce the oracle's (= the reference's) quantised coefficients exactly."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import MANIFEST, ROOT, SMALL_CASES, golden_jpeg
from libjpeg_amd import api, synth


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mijpeg.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)  # declarations only, no comments
    names = sorted(set(re.findall(r"\b(mijpeg_[a-z_]+)\s*\(", hdr)))
    assert len(names) >= 15
    L = ctypes.CDLL(api.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/mijpeg.h but not exported"


def test_info_struct_layout_matches_header(tmp_path):
    """The ctypes mirrors of libjpeg_amd/api.py against include/mijpeg.h as the C compiler lays it out: sizes and the offsets of
    the last members."""
    import subprocess
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mijpeg.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(mijpeg_info), offsetof(mijpeg_info, coef_wide), '
                   'offsetof(mijpeg_info, coef_count), sizeof(mijpeg_xt_params), offsetof(mijpeg_xt_params, r2table), sizeof(mijpeg_batch)); return 0; }\n')
    exe = tmp_path / "layout"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), "-o", str(exe), str(src)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert got == [ctypes.sizeof(api.MijpegInfo), api.MijpegInfo.coef_wide.offset, api.MijpegInfo.coef_count.offset,
                   ctypes.sizeof(api.MijpegXtParams), api.MijpegXtParams.r2table.offset, ctypes.sizeof(api.MijpegBatch)]


@pytest.mark.parametrize("name", SMALL_CASES)
def test_headers_match_oracle(oracle, name):
    data = golden_jpeg(name)
    d = api.Decoder(None)
    f = d.read_header(data)
    o = oracle.read_info(data)
    assert (f.width, f.height, f.components) == (o.width, o.height, o.ncomp)
    for c in range(o.ncomp):
        assert (f.hsamp[c], f.vsamp[c], f.subx[c], f.suby[c]) == (o.hs[c], o.vs[c], o.subx[c], o.suby[c])
        assert (f.blocks_w[c], f.blocks_h[c]) == (o.bw[c], o.bh[c])
        assert list(f.quant[f.quant_index[c]]) == list(o.quant[o.tq[c]])
    assert f.ycbcr == o.ycbcr
    d.close()


@pytest.mark.parametrize("threads", [1, 3])
@pytest.mark.parametrize("name", SMALL_CASES)
def test_coefficients_match_oracle(oracle, name, threads):
    data = golden_jpeg(name)
    d = api.Decoder(None)
    f = d.read(data, threads)
    _, planes = oracle.decode_coefficients(data)
    for c in range(f.components):
        assert np.array_equal(d.coefficients(c), planes[c].astype(np.int16)), f"component {c}"
    assert f.fast_arith == 1  # every real image passes the range check
    d.close()


@pytest.mark.parametrize("w,h,sub,dri", [(640, 360, "420", 8), (641, 363, "420", 5), (500, 300, "444", 1), (512, 512, "422", 0)])
def test_coefficients_larger_streams(oracle, w, h, sub, dri):
    data = synth.synth_jpeg(w, h, 77, 85, sub, dri)
    d = api.Decoder(None)
    f = d.read(data, 4)
    _, planes = oracle.decode_coefficients(data)
    for c in range(f.components):
        assert np.array_equal(d.coefficients(c), planes[c].astype(np.int16))
    d.close()


@pytest.mark.parametrize("dri,progressive", [(1, False), (8, False), (0, False), (8, True)])
def test_parallel_marker_search_on_a_stream_of_several_chunks(oracle, dri, progressive):
    """Streams beyond 1 MiB have their restart markers located by parallel chunks (HostDecoder::find_intervals)."""
    data = synth.encode_jpeg(synth.synth_image(2560, 1440, 5), 95, "444", restart_mcus=dri, progressive=progressive)
    assert len(data) > (1 << 20)
    d = api.Decoder(None)
    f = d.read(data, 4)
    _, planes = oracle.decode_coefficients(data)
    for c in range(f.components):
        assert np.array_equal(d.coefficients(c), planes[c].astype(np.int16))
    d.close()


@pytest.mark.parametrize("dri", [8, 0])
def test_segment_ends_of_a_large_progressive_file_come_from_one_pass_of_the_pool(oracle, dri):
    """Beyond 2 MiB the planning walk over a progressive frame asks HostDecoder::segment_end, which has the pool note every
    segment-ending marker of the file once and answers every scan from that list."""
    data = synth.encode_jpeg(synth.synth_image(3200, 2000, 11), 97, "444", restart_mcus=dri, progressive=True)
    assert len(data) > (5 << 19)
    d = api.Decoder(None)
    f = d.read(data, 4)
    _, planes = oracle.decode_coefficients(data)
    for c in range(f.components):
        assert np.array_equal(d.coefficients(c), planes[c].astype(np.int16))
    d.close()


@pytest.mark.parametrize("w,h,sub,q", [(2560, 1440, "444", 95), (3840, 2160, "420", 85), (1600, 1200, "422", 90), (2048, 2048, "gray", 92)])
def test_streams_without_restart_markers_decode_in_parallel(oracle, w, h, sub, q):
    """No DRI: the entropy coded segment is cut into ranges that are decoded speculatively and stitched where they
    synchronise (HostDecoder::decode_scan_speculative); the coefficients must be the sequential decoder's."""
    img = synth.synth_image(w, h, 9, channels=1 if sub == "gray" else 3)
    data = synth.encode_jpeg(img, q, sub if sub != "gray" else "444")
    d = api.Decoder(None)
    before = api.speculative_scans()[0]
    f = d.read(data, 8)
    assert api.speculative_scans()[0] == before + 1  # really taken, not the fallback
    one = api.Decoder(None)
    one.read(data, 1)  # one thread: the plain sequential path
    assert api.speculative_scans()[0] == before + 1
    for c in range(f.components):
        assert np.array_equal(d.coefficients(c), one.coefficients(c)), c
    assert list(d.info.range_max) == list(one.info.range_max)
    if w * h <= 2560 * 1440:
        _, planes = oracle.decode_coefficients(data)
        for c in range(f.components):
            assert np.array_equal(d.coefficients(c), planes[c].astype(np.int16))
    d.close()
    one.close()


def test_speculative_decoder_with_tiny_ranges():
    """Ranges of a few dozen bytes (hundreds of synchronisation points, ranges that never synchronise, ranges cut at
    stuffed bytes) on every small sequential stream without restart markers -- in a subprocess because the range
    size is an environment setting."""
    import subprocess
    import sys

    code = """
import sys, numpy as np
sys.path.insert(0, %r)
from tests.conftest import SMALL_CASES, MANIFEST, golden_jpeg
from libjpeg_amd import api, synth
n = 0
streams = [golden_jpeg(k) for k in SMALL_CASES if not MANIFEST[k].get("dri") and "-z" not in (MANIFEST[k].get("args") or []) and "prog" not in k]
rng = np.random.default_rng(1)
for q in (30, 75, 98):
    streams.append(synth.encode_jpeg(rng.integers(0, 256, (96, 160, 3)).astype(np.uint8), q, "420"))  # noise: many 0xFF bytes
for data in streams:
    a, b = api.Decoder(None), api.Decoder(None)
    fa = a.read(data, 7)
    fb = b.read(data, 1)
    for c in range(fa.components):
        assert np.array_equal(a.coefficients(c), b.coefficients(c))
    assert list(fa.range_max) == list(fb.range_max)
    n += 1
scans, pieces = api.speculative_scans()
assert scans >= 1 and pieces >= 4 * scans, (scans, pieces, n)  # (streams shorter than four ranges take the sequential path)
print("checked", n)
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    for seg in ("64", "333", "4096"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, MIJPEG_SPEC_SEGMENT_BYTES=seg))
        assert r.returncode == 0, r.stderr[-2000:]
        assert "checked" in r.stdout


def test_kernel_selection_and_workspace_need_no_device():
    """mijpeg_kernel_name / mijpeg_workspace_bytes are host logic: the fused kernel per layout, and frames x 1 KiB more
    scratch when the frames of a batch bring their own quantisation tables (mijpeg_batch.quant_dev)."""
    d = api.Decoder(None)
    expect = {"ref_80x48_420": "fused420p_kernel", "ref_100x9_422": "fused422_kernel", "ref_97x61_440": "fused440_kernel"}
    for name, kernel in expect.items():
        f = d.read(golden_jpeg(name))
        assert f.fast_arith == 1
        assert api.kernel_name(f) == kernel
        assert api.kernel_name(f, api.FLAG_FORCE_SAFE) == ("fused420_kernel" if "420" in name else "fused_tile_kernel")
        assert api.kernel_name(f, api.FLAG_FORCE_GENERIC) == "idct_planes_kernel+upsample_color_kernel"
        assert api.workspace_bytes(f, 5) == 0
        assert api.workspace_bytes(f, 5, own_tables=True) == 5 * 4 * 64 * 4
        generic = api.workspace_bytes(f, 5, api.FLAG_FORCE_GENERIC)
        assert generic >= 5 * 4 * int(f.coef_count)
        assert api.workspace_bytes(f, 5, api.FLAG_FORCE_GENERIC, own_tables=True) == generic + 5 * 4 * 64 * 4
    d.close()


def test_reconstruct_without_device_fails_loudly():
    d = api.Decoder(None)
    d.read(golden_jpeg("ref_80x48_420"))
    with pytest.raises(api.MijpegError) as e:
        d.reconstruct()
    assert e.value.code == api.ERR_DEVICE
    d.close()


def _coefficients_match_the_oracle(oracle, data):
    d = api.Decoder(None)
    d.read(data, entropy="host")
    info, planes = oracle.decode_coefficients(data)
    for c in range(info.ncomp):
        assert np.array_equal(d.coefficients(c), planes[c]), c
    d.close()


def test_error_codes_are_the_references(oracle):
    d = api.Decoder(None)
    with pytest.raises(api.MijpegError) as e:
        d.read(b"not a jpeg at all")
    assert e.value.code == -1038  # codestream/decoder.cpp:94-96: MALFORMED_STREAM, "SOI marker missing"
    data = golden_jpeg("ref_80x48_420")
    # lossless frame (SOF3) -> NOT_IMPLEMENTED on this path; sequential scan parameters in a progressive frame -> malformed
    with pytest.raises(api.MijpegError) as e:
        d.read(data.replace(b"\xff\xc0", b"\xff\xc3", 1))
    assert e.value.code == -1034
    with pytest.raises(api.MijpegError) as e:
        d.read(data.replace(b"\xff\xc0", b"\xff\xc2", 1))
    assert e.value.code == -1038
    d.close()
    # a stream without restart markers that simply ends: the reference's bit reader pads with zero bits at the end of the
    # data (io/bitstream.cpp:96-101) and a missing EOI only warns (marker/frame.cpp:1099-1102): it decodes, and so do we
    _coefficients_match_the_oracle(oracle, data[: len(data) // 2])
    # with restart markers the entropy parser runs into the end while looking for the next one (entropyparser.cpp:141-146)
    data = golden_jpeg("pil_200x120_420_dri8")
    d = api.Decoder(None)
    with pytest.raises(api.MijpegError) as e:
        d.read(data[: len(data) // 2], entropy="host")
    assert e.value.code == -1025
    d.close()


def test_restart_markers_out_of_sequence_resynchronise(oracle):
    # codestream/entropyparser.cpp:117-201: RST1 renumbered to RST3 looks like a marker two intervals ahead -- the
    # decoder gives up the intervals in between (grey) and is in step again where the numbering agrees
    data = bytearray(golden_jpeg("pil_200x120_420_dri8"))
    i = data.index(b"\xff\xd1")
    data[i + 1] = 0xD3
    _coefficients_match_the_oracle(oracle, bytes(data))
    # a dropped interval (marker and data): one grey interval, everything else in place
    data = golden_jpeg("pil_200x120_420_dri8")
    a, b = data.index(b"\xff\xd2"), data.index(b"\xff\xd3")
    _coefficients_match_the_oracle(oracle, data[:a] + data[b:])


def test_range_check_rejects_absurd_coefficients(oracle):
    # DC-only stream with a huge quantiser: sum |c| q can exceed the fast-arithmetic bound
    data = bytearray(golden_jpeg("ref_64x40_q2"))
    d = api.Decoder(None)
    f = d.read(bytes(data))
    # q = 2 tables have deltas up to 255*...: just check the flag is a 0/1 decision consistent with the bound
    worst = 0
    for c in range(f.components):
        q = np.array(f.quant[f.quant_index[c]][:], np.int64)
        worst = max(worst, int((np.abs(d.coefficients(c).astype(np.int64)) * q).sum(axis=2).max()))
    assert f.fast_arith == (1 if worst < 16384 else 0)
    assert max(f.range_max[c] for c in range(f.components)) == worst
    d.close()


def test_mutated_streams_never_crash():
    """Robustness of the host side: byte flips, truncations and insertions in valid streams (sequential, progressive,
    12 bit, JPEG XT) must end in a decoded frame or a MijpegError -- never in a crash or a hang."""
    from conftest import P12_CASES, XT_CASES

    rng = np.random.default_rng(12345)
    names = ["pil_200x120_420_dri8", "ref_75x45_420_dri2", "pilprog_75x45_420", "refprog_64x64_444_dri5", "pil_70x40_gray",
             XT_CASES[0], P12_CASES[0]]
    d = api.Decoder(None)
    outcomes = {"ok": 0, "error": 0}
    for name in names:
        base = bytearray(golden_jpeg(name))
        for trial in range(150):
            data = bytearray(base)
            kind = trial % 4
            if kind == 0:  # flip a few bytes anywhere
                for _ in range(int(rng.integers(1, 6))):
                    data[int(rng.integers(2, len(data)))] = int(rng.integers(0, 256))
            elif kind == 1:  # damage the headers specifically
                for _ in range(int(rng.integers(1, 4))):
                    data[int(rng.integers(2, min(len(data), 700)))] = int(rng.integers(0, 256))
            elif kind == 2:  # truncate
                data = data[: int(rng.integers(4, len(data)))]
            else:  # insert garbage
                at = int(rng.integers(2, len(data)))
                data[at:at] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 40)), dtype=np.uint8))
            try:
                d.read(bytes(data), 2)
                outcomes["ok"] += 1
            except api.MijpegError as e:
                assert e.code < 0
                outcomes["error"] += 1
    d.close()
    assert outcomes["error"] > 100 and outcomes["ok"] + outcomes["error"] == 150 * len(names)


def test_host_decoder_under_thread_sanitizer():
    """tools/tsan_host.sh: the restart-parallel and the speculative decode on the shared worker pool, and several decoder
    objects decoding (also damaged) streams from different caller threads, under -fsanitize=thread: no data race, and the
    8-thread decode equals the 1-thread decode."""
    import os
    import shutil
    import subprocess

    from conftest import ROOT

    if not shutil.which("g++"):
        pytest.skip("no g++")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "tsan_host.sh")], capture_output=True, text=True, timeout=600)
    if "unsupported option" in r.stderr or "cannot find -ltsan" in r.stderr or "libtsan" in r.stderr and r.returncode != 0 and "WARNING" not in r.stderr:
        pytest.skip("this toolchain has no ThreadSanitizer runtime")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "mismatched 0" in r.stdout and "ThreadSanitizer" not in r.stderr


def _unstuffed_reference(data: bytes):
    """Python restatement: entropy coded data of the first scan, per restart interval the bytes up to the first FF that is not
    followed by 00, FF 00 as FF."""
    import damage
    es = damage.entropy_start(data)
    out, begins, cur, p = bytearray(), [0], bytearray(), es
    stopped = False
    while p < len(data):
        b = data[p]
        if b != 0xFF:
            if not stopped:
                cur.append(b)
            p += 1
            continue
        nxt = data[p + 1] if p + 1 < len(data) else None
        if nxt == 0x00:
            if not stopped:
                cur.append(0xFF)
            p += 2
        elif nxt == 0xFF:
            stopped = True
            p += 1
        elif nxt is not None and 0xD0 <= nxt <= 0xD7:
            out += cur
            begins.append(len(out))
            cur, stopped = bytearray(), False
            p += 2
        else:
            break
    out += cur
    return bytes(out), begins


def test_unstuffed_scan_matches_a_python_restatement():
    """What the device decoder is fed (mijpeg_unstuffed_scan): for streams with and without restart markers, noise (many FF
    bytes), fill bytes in front of markers, and -- in a subprocess with tiny search chunks and tiny copy pieces -- every seam
    between the parallel pieces, stuffed pairs that straddle one included."""
    import subprocess
    import sys
    code = """
import sys, os, numpy as np, ctypes as C
sys.path.insert(0, %r); sys.path.insert(0, %r)
from libjpeg_amd import api, synth
from test_host_decoder import _unstuffed_reference
import damage
rng = np.random.default_rng(3)
streams = []
for dri in (0, 1, 3, 8):
    streams.append(synth.encode_jpeg(rng.integers(0, 256, (72, 104, 3)).astype(np.uint8), 97, "420", restart_mcus=dri))   # noise
    streams.append(synth.synth_jpeg(200, 120, 5 + dri, 85, "444", dri))
# fill bytes in front of the restart markers and of EOI
s = bytearray(streams[2]); out = bytearray(); i = 0
es = damage.entropy_start(bytes(s))
while i < len(s):
    if i >= es and s[i] == 0xFF and i + 1 < len(s) and (0xD0 <= s[i + 1] <= 0xD9):
        out += b"\\xff\\xff"
    out.append(s[i]); i += 1
streams.append(bytes(out))
L = api.lib()
L.mijpeg_unstuffed_scan.restype = C.c_int64
L.mijpeg_unstuffed_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_int32)]
n = 0
for data in streams:
    exp, begins = _unstuffed_reference(data)
    d = api.Decoder(None)
    d.read(data)
    for piece in (16, 100, 1 << 20):
        nint = C.c_int32(0)
        total = L.mijpeg_unstuffed_scan(d._h, None, 0, None, 0, piece, C.byref(nint))
        assert total == len(exp), (total, len(exp))
        assert nint.value == len(begins)
        buf = (C.c_uint8 * (total + 16))(*([0xAA] * (total + 16)))
        b = (C.c_uint32 * nint.value)()
        assert L.mijpeg_unstuffed_scan(d._h, buf, total, b, nint.value, piece, None) == total
        assert bytes(buf[:total]) == exp, piece
        assert bytes(buf[total:]) == b"\\xaa" * 16
        assert list(b) == begins
    # the other producer: the marker search writes the copy as it goes (batches)
    big = (C.c_uint8 * (len(data) + 64))()
    got = L.mijpeg_unstuffed_scan(d._h, big, len(data) + 64, b, nint.value, 1, None)
    if os.environ.get("MIJPEG_MARKER_CHUNK"):
        assert got == -1029  # (tiny search chunks: the search runs in parallel pieces and leaves the copy to the gather)
    else:
        assert got == len(exp) and bytes(big[:got]) == exp and list(b) == begins
    d.close()
    n += 1
print("checked", n)
""" % (ROOT, os.path.join(ROOT, "tests"))
    for chunk in ("0", "16", "61"):
        env = dict(os.environ)
        if chunk != "0":
            env["MIJPEG_MARKER_CHUNK"] = chunk
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "checked 9" in r.stdout


def test_progressive_scans_searched_in_tiny_chunks_give_the_same_coefficients(oracle):
    """The restart marker searches of a progressive frame's scans run as one list of chunk tasks (run_deferred_searches); with
    MIJPEG_MARKER_CHUNK the chunks are a few bytes long and every seam -- a marker, a stuffed pair, a fill byte astride two
    chunks -- is met: same coefficient planes as with the default chunks, which are the oracle's."""
    import subprocess
    import sys
    code = """
import sys, os, hashlib, numpy as np
sys.path.insert(0, %r)
from libjpeg_amd import api, synth
rng = np.random.default_rng(5)
for dri in (1, 3, 8):
    for img in (rng.integers(0, 256, (72, 104, 3)).astype(np.uint8), synth.synth_image(200, 120, 5 + dri)):
        data = synth.encode_jpeg(img, 95, "420", restart_mcus=dri, progressive=True)
        d = api.Decoder(None)
        f = d.read(data, 4)
        print(hashlib.sha256(b"".join(d.coefficients(c).tobytes() for c in range(f.components))).hexdigest())
        d.close()
""" % (ROOT,)
    outs = []
    for chunk in ("0", "16", "61"):
        env = dict(os.environ)
        if chunk != "0":
            env["MIJPEG_MARKER_CHUNK"] = chunk
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.split())
    assert len(outs[0]) == 6 and outs[0] == outs[1] == outs[2]
    # ... and the default chunks' planes are the oracle's
    rng = np.random.default_rng(5)
    data = synth.encode_jpeg(rng.integers(0, 256, (72, 104, 3)).astype(np.uint8), 95, "420", restart_mcus=1, progressive=True)
    d = api.Decoder(None)
    f = d.read(data, 4)
    _, planes = oracle.decode_coefficients(data)
    for c in range(f.components):
        assert np.array_equal(d.coefficients(c), planes[c].astype(np.int16))
    d.close()


def test_scan_offsets_are_where_the_scan_headers_end():
    """mijpeg_scan_offsets (what JPGFLAG_DECODER_STOP_SCAN returns at): every SOS of the codestream, first entropy coded byte
    behind its header, and the marker that follows the scan's data; JPEG XT scans from boxes come last with end 0."""
    import ctypes as C
    L = api.lib()
    L.mijpeg_scan_offsets.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]
    for name in ("pilprog_75x45_420", "pil_200x120_420_dri8", "xt_129x71_420_rR2"):
        data = golden_jpeg(name)
        d = api.Decoder(None)
        d.read(data)
        first, end = (C.c_uint64 * 64)(), (C.c_uint64 * 64)()
        n = L.mijpeg_scan_offsets(d._h, first, end, 64)
        sos = []
        p = 2
        while p + 4 <= len(data):  # walk the marker segments and the entropy coded data between them
            assert data[p] == 0xFF
            m = data[p + 1]
            if m == 0xD9:
                break
            ln = (data[p + 2] << 8) | data[p + 3]
            p += 2 + ln
            if m == 0xDA:
                sos.append(p)
                while not (data[p] == 0xFF and data[p + 1] != 0 and not 0xD0 <= data[p + 1] <= 0xD7):
                    p += 1
        main = [k for k in range(n) if end[k] != 0]
        assert [first[k] for k in main] == sos, name
        for k in main:
            assert data[end[k]] == 0xFF and data[end[k] + 1] not in (0,) and not 0xD0 <= data[end[k] + 1] <= 0xD7
        if name.startswith("xt_"):
            assert n > len(main) and all(first[k] == end[main[-1]] for k in range(n) if end[k] == 0)
        else:
            assert n == len(main)
        d.close()


def test_prepare_batch_host_is_the_host_half_of_a_batch_submit():
    """mijpeg_prepare_batch_host on a host-only object: header parse + marker search + unstuffed copy of every stream, the
    reference's error for a stream that is none."""
    from libjpeg_amd import synth
    streams = [synth.synth_jpeg(160 + 8 * i, 96, i, 85, "420", 4) for i in range(6)]
    d = api.Decoder(None)
    d.prepare_batch_host(streams)
    d.prepare_batch_host(streams[:2])
    with pytest.raises(api.MijpegError):
        d.prepare_batch_host([streams[0], b"\xff\xd8 not a jpeg at all"])
    d.close()


def test_pipelined_scans_leave_the_same_coefficient_store():
    """Progressive frames and JPEG XT frames with hidden refinement scans: the scans without restart intervals run side by side,
    a scan trailing the one it refines by an MCU row, and a residual frame decodes while the calling thread takes the legacy
    frame.  The whole coefficient store (legacy and residual planes) must be what scan after scan, frame after frame leaves
    (MIJPEG_NO_SCAN_PIPELINE / MIJPEG_NO_FRAME_OVERLAP, read once per process: two child processes), at 1, 4 and 16 threads."""
    import glob
    import subprocess
    import sys
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "tests", "golden", "*.jpg")) if "prog" in os.path.basename(f) or "xt_" in os.path.basename(f))
    assert len(files) >= 20
    code = r"""
import sys, ctypes as C, hashlib
sys.path.insert(0, %r)
from libjpeg_amd import api
L = api.lib()
L.mijpeg_coefficients.restype = C.c_void_p
for path in sys.argv[1:]:
    data = open(path, "rb").read()
    for thr in (1, 4, 16):
        d = api.Decoder(None)
        f = d.read(data, thr)
        base = L.mijpeg_coefficients(d._h, 0) - int(f.coef_offset[0]) * 2
        n = int(f.coef_count) * (4 if f.coef_wide else 2)
        rr = list(d.xt_params().residual.range_max)[:3] if f.xt else []  # (the residual frame's: its last window takes it on the way)
        print(path, thr, hashlib.sha256(bytes((C.c_uint8 * n).from_address(base))).hexdigest(), list(f.range_max)[:f.components], rr)
        d.close()
""" % ROOT
    outs = []
    # (third child: the pipeline with the blocks themselves as the medium between the scans of a refinement chain, not bit masks)
    for extra in ({}, {"MIJPEG_NO_SCAN_PIPELINE": "1", "MIJPEG_NO_FRAME_OVERLAP": "1"}, {"MIJPEG_NO_DEFERRED_REFINE": "1"}):
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, "-c", code] + files, capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.splitlines())
    assert len(outs[0]) == 3 * len(files) and outs[0] == outs[1] == outs[2]
    # ... and the thread count changes nothing either
    for i in range(0, len(outs[0]), 3):
        assert len({" ".join(ln.split()[2:]) for ln in outs[0][i:i + 3]}) == 1, outs[0][i:i + 3]


def test_scan_pipeline_honours_the_callers_thread_bound():
    """A JPEG XT stream with hidden refinement scans (-R 3 -rR 4: 3 + 16 scans in boxes) decoded with threads = 2 in a fresh process: the
    worker pool grows by the two threads asked for (plus the one that drives the residual frame), not by one per scan; the
    coefficients are those of the unbounded decode."""
    import subprocess
    import sys
    code = r'''
import os, sys, hashlib
sys.path.insert(0, %r)
from libjpeg_amd import api
data = open(%r, "rb").read()
before = len(os.listdir("/proc/self/task"))
d = api.Decoder(None)
f = d.read(data, threads=2)
after = len(os.listdir("/proc/self/task"))
h = hashlib.sha256(b"".join(d.coefficients(c).tobytes() for c in range(f.components))).hexdigest()
d.close()
d = api.Decoder(None)
f = d.read(data, threads=16)
h16 = hashlib.sha256(b"".join(d.coefficients(c).tobytes() for c in range(f.components))).hexdigest()
print(after - before, h == h16)
''' % (ROOT, os.path.join(ROOT, "tests", "golden", "xt_200x120_420_R3_rR4.jpg"))
    out = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) <= 3 and out[1] == "True", out


def test_mask_based_refinement_pass_equals_the_plain_one():
    """The AC refinement pass that keeps a block's history as bit masks (BMI2 / AVX2, picked at run time) and the parallel first
    full-band pass against the position-by-position decoder (MIJPEG_NO_REFINE_MASKS / MIJPEG_NO_SPEC_FIRST_PASS, a process of its
    own): the same coefficients on progressive pictures, JPEG XT frames with hidden refinement scans and damaged versions of both."""
    import subprocess
    import sys
    code = r'''
import sys, hashlib
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from libjpeg_amd import api
import damage
from conftest import MANIFEST, golden_jpeg
names = sorted(n for n in MANIFEST if not MANIFEST[n].get("big") and ("prog" in n or "R" in n.split("_")[-1] or "_R" in n or "rR" in n))
d = api.Decoder(None)
out = []
for fi, n in enumerate(names):
    blobs = [golden_jpeg(n)] + [b for _, b in damage.cases(golden_jpeg(n), 6, 7000 + fi, "entropy")]
    for blob in blobs:
        for th in (1, 5, 16):  # (16: the refinement chains of a component run in one window, on bit masks passed along)
            try:
                f = d.read(blob, threads=th)
                h = hashlib.sha256(b"".join(d.coefficients(c).tobytes() for c in range(f.components)))
                if f.xt:
                    x = d.xt_params()
                    h.update(bytes(str((x.hidden_bits, x.residual_hidden_bits)), "ascii"))
                    for c in range(3):
                        h.update(d.residual_coefficients(c).tobytes())
                out.append(h.hexdigest()[:16])
            except api.MijpegError as e:
                out.append(str(e.code))
print(len(names), hashlib.sha256(" ".join(out).encode()).hexdigest())
''' % (ROOT, os.path.join(ROOT, "tests"))
    res = []
    for env in ({}, {"MIJPEG_NO_REFINE_MASKS": "1", "MIJPEG_NO_SPEC_FIRST_PASS": "1"}, {"MIJPEG_NO_DEFERRED_REFINE": "1"}, {"MIJPEG_NO_TRAILING_APPLY": "1"}):
        r = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True, env=dict(os.environ, **env))
        res.append(r.stdout.split())
    assert res[0] == res[1] == res[2] == res[3] and int(res[0][0]) >= 12, res
