"""Damaged JPEG XT streams: same input, same result as the reference.

What is specific to JPEG XT: the reference turns to a frame's hidden refinement scans only when the frame trailer stands at an
EOI marker (marker/frame.cpp:1063-1070) and to the residual codestream only when the image trailer does
(codestream/image.cpp:1416-1431).  A legacy codestream that runs out without one -- truncated, or with a marker where none
belongs -- is reconstructed WITHOUT the residual: the legacy samples through the L tables alone (rr = m_lOutDCShift,
colortrafo/ycbcrtrafo.cpp:744-746).  What is wrong with the residual codestream's header (dimensions, a DNL marker) is reported
behind the legacy frame's decode, whose own errors come first.  Damage inside the entropy coded data of either codestream goes
through the reference's resynchronisation like in a plain JPEG (tests/test_damaged.py).
Layers: oracle against the reference binary live (build container), product host decoder against the oracle (error codes),
-m gpu: product pixels (16-bit codes) against the oracle's.
"""
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest

import damage
from conftest import MANIFEST, XT_CASES, golden_jpeg
from libjpeg_amd import api


def handmade():
    """name -> stream: truncations and deletions of whole scans / boxes."""
    out = {}
    for name in ("xt_129x71_420", "xt_129x71_420_R2_rR3_dri3", "xt_64x48_444_R1_rR1", "xt_200x120_420_rR4"):
        data = golden_jpeg(name)
        out[name + "/no_eoi"] = data[:-2]
        out[name + "/cut_90"] = data[:len(data) * 9 // 10]
        out[name + "/cut_legacy_scan"] = data[:damage.entropy_start(data) + 200]
        # the last APP11 segment in front of the legacy frame's scan (a piece of the residual codestream or a refinement box)
        segs = [m.start() for m in re.finditer(b"\xff\xeb", data[:damage.entropy_start(data)])]
        if len(segs) > 4:
            a = segs[-2]
            ln = (data[a + 2] << 8) | data[a + 3]
            out[name + "/drop_box_segment"] = data[:a] + data[a + 2 + ln:]
    return out


HANDMADE = handmade()


def seeded(per_file, seed):
    for fi, name in enumerate(XT_CASES):
        data = golden_jpeg(name)
        for where in ("any", "entropy"):
            for kind, blob in damage.cases(data, per_file, seed * 1000 + fi + (500 if where == "entropy" else 0), where):
                yield name, kind, blob


def reference_status(oracle, data):
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as d:
        src, dst = os.path.join(d, "in.jpg"), os.path.join(d, "out.pfm")
        with open(src, "wb") as f:
            f.write(data)
        try:
            r = subprocess.run([oracle.REF_BIN, src, dst], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=20)
        except subprocess.TimeoutExpired:
            return None, "timeout"
        if r.returncode < 0:
            return None, "crash"
        m = re.search(rb"failed - error (-?\d+)", r.stderr)
        if m:
            return None, int(m.group(1))
        try:
            return oracle.read_pfm_reference(dst), 0
        except Exception:  # noqa: BLE001
            return None, "no output"


def test_oracle_against_live_reference_on_damaged_xt_streams(oracle):
    if not oracle.have_reference():
        pytest.skip("oracle/_ref/jpeg not built")
    from concurrent.futures import ThreadPoolExecutor
    work = [(k, "handmade", v) for k, v in HANDMADE.items()] + list(seeded(7, 51))

    def one(item):
        name, kind, blob = item
        codes, is_float, oerr = oracle.decode_xt_status(blob)
        if oerr is None:
            return None
        rpx, rerr = reference_status(oracle, blob)
        if rerr in ("timeout", "crash"):
            return None
        if rerr != oerr:
            return f"{name}/{kind}: reference {rerr}, oracle {oerr}"
        if rerr == 0:
            opx = oracle.half_codes_to_float(codes)
            if rpx.shape != opx.shape or not np.array_equal(rpx.view(np.uint32), opx.view(np.uint32)):
                return f"{name}/{kind}: pixels differ"
        return ""

    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(one, work))
    bad = [r for r in res if r]
    assert not bad, bad[:10]
    assert sum(r == "" for r in res) >= 150
    # the handmade truncations decode -- without their residual
    for k in HANDMADE:
        if k.endswith("/no_eoi"):
            codes, _, err = oracle.decode_xt_status(HANDMADE[k])
            full, _, _ = oracle.decode_xt_status(golden_jpeg(k.split("/")[0]))
            assert err == 0 and not np.array_equal(codes, full), k


def test_host_decoder_verdicts_on_damaged_xt_streams(oracle):
    """Return codes of the product's host decoder == the oracle's (= the reference's) on the same corruptions."""
    d = api.Decoder(None)
    n = 0
    bad = []
    for name, kind, blob in [(k, "handmade", v) for k, v in HANDMADE.items()] + list(seeded(7, 52)):
        _, _, oerr = oracle.decode_xt_status(blob)
        if oerr is None:
            continue
        try:
            d.read(blob)
            perr = 0
        except api.MijpegError as e:
            perr = e.code
        if perr != oerr and not (perr == -1034 and oerr == 0):  # (-1034: a subset the accelerated path declines, never a wrong picture)
            bad.append((name, kind, oerr, perr))
        n += 1
    d.close()
    assert not bad, bad[:10]
    assert n >= 150


@pytest.mark.gpu
def test_gpu_damaged_xt_pixels(oracle):
    dec = api.Decoder(0)
    stats = {"ok": 0, "declined": 0}
    bad = []
    for name, kind, blob in [(k, "handmade", v) for k, v in HANDMADE.items()] + list(seeded(5, 53)):
        codes, _, oerr = oracle.decode_xt_status(blob)
        if oerr is None:
            continue
        try:
            dec.read(blob)
            perr = 0
        except api.MijpegError as e:
            perr = e.code
        if perr == -1034 and oerr == 0:
            stats["declined"] += 1
            continue
        if perr != oerr:
            bad.append((name, kind, oerr, perr))
            continue
        if perr == 0:
            out = dec.reconstruct()
            if out.shape != codes.shape or not np.array_equal(out, codes):
                bad.append((name, kind, "pixels", int(np.count_nonzero(out != codes)) if out.shape == codes.shape else -1))
                continue
        stats["ok"] += 1
    dec.close()
    print(stats)
    assert not bad, bad[:10]
    assert stats["ok"] >= 150 and stats["declined"] <= 10


def test_the_residual_images_quantiser_tables_are_looked_up_at_the_first_request(oracle):
    """The RESIDUAL image's transforms are built when its first block is dequantised (ResidualBlockHelper::AllocateBuffers ->
    Tables::BuildDCT -> FindQuantizationTable, codestream/tables.cpp:1481-1494, 1750), not at the start of its first scan like the
    legacy image's: a residual codestream without its DQT marker is -1031 -- but whatever stops its entropy coded data comes first
    (tools/box_campaign.py r5: a DQT marker turned into garbage AND a damaged scan, -1038 in the reference)."""
    data = golden_jpeg("xt_129x71_420")
    i = data.index(b"RESI") + 4
    assert data[i:i + 4] == b"\xff\xd8\xff\xdb"
    no_dqt = data[:i + 2] + b"\x5e" + data[i + 3:]  # FF DB -> 5E DB: the parser skips to the next marker, the table never arrives
    rsos = no_dqt.index(b"\xff\xda", i)
    both = bytearray(no_dqt)
    both[rsos + 14 + 60:rsos + 14 + 62] = b"\xff\xc4"  # a DHT marker in the residual scan's data
    both = bytes(both)
    for blob, want in ((no_dqt, -1031), (both, None)):
        oerr = oracle.decode_xt_status(blob)[2]
        if oracle.have_reference():
            rerr = reference_status(oracle, blob)[1]
            assert rerr == oerr, (rerr, oerr)
        if want is not None:
            assert oerr == want
        else:
            assert oerr not in (0, -1031)
        d = api.Decoder(None)
        with pytest.raises(api.MijpegError) as e:
            d.read(blob)
        assert e.value.code == oerr
        d.close()


def _residual_without_its_visible_scan():
    """tests/golden/xt_grey/ghdr_R1_rR3 with the SOS marker of its residual codestream turned into a reserved marker (FF DA -> FF 58):
    the residual frame's component is named by no visible scan and appears in the hidden refinement scans of the RFIN boxes first
    (found by tools/xt_gpu_damage_campaign.py, seed 72)."""
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "xt_grey", "ghdr_R1_rR3.jpg"), "rb") as f:
        data = f.read()
    i = data.index(b"RESI")
    sos = data.index(b"\xff\xda", i)
    assert sos < data.index(b"FINE", i)  # (still inside the RESI box)
    return data, data[:sos + 1] + b"\x58" + data[sos + 2:]


def test_a_component_that_first_appears_in_a_hidden_scan_takes_the_table_in_force_there(oracle):
    """The transform of a component is built, with the quantiser table in force then, when the component first appears in a scan
    (control/blockbuffer.cpp:177-208) -- for a residual frame whose scan header is gone that is the first hidden refinement scan;
    the product published the stand-in of a component that appears nowhere (a table of ones) before it had looked at the boxes."""
    good, blob = _residual_without_its_visible_scan()
    codes, _, oerr = oracle.decode_xt_status(blob)
    assert oerr == 0
    if oracle.have_reference():
        ref = oracle.reference_decode_hdr(blob)
        assert np.array_equal(ref.reshape(-1), oracle.half_codes_to_float(np.asarray(codes)).reshape(-1).astype(np.float32))
    tables = []
    for data in (good, blob):
        d = api.Decoder(None)
        d.read(data)
        r = d.xt_params().residual
        tables.append(np.ctypeslib.as_array(r.quant).reshape(4, 64)[list(r.quant_index)[0]].copy())
        d.close()
    assert np.array_equal(tables[0], tables[1]) and tables[0].max() > 1


@pytest.mark.gpu
def test_gpu_residual_frame_without_its_visible_scan(oracle):
    _, blob = _residual_without_its_visible_scan()
    codes, _, oerr = oracle.decode_xt_status(blob)
    assert oerr == 0
    dec = api.Decoder(0)
    dec.read(blob)
    out = dec.reconstruct()
    dec.close()
    assert out.shape == codes.shape and np.array_equal(out, codes)


def _residual_without_a_specification():
    """A RESI box and no merging specification (the SPEC box's type damaged): the reference reads the legacy frame like a plain
    JPEG's and refuses at the first request -- -1024, "The combination of L and R transformation is non-standard"."""
    data = golden_jpeg("xt_129x71_420_rR1_dri2")
    i = data.index(b"SPEC")
    return data[:i] + b"SPEX" + data[i + 4:]


def test_a_residual_codestream_without_a_specification_is_refused_behind_the_legacy_frame(oracle):
    blob = _residual_without_a_specification()
    assert oracle.decode_xt_status(blob)[2] == -1024
    if oracle.have_reference():
        assert reference_status(oracle, blob)[1] == -1024
    d = api.Decoder(None)
    with pytest.raises(api.MijpegError) as e:
        d.read(blob)
    assert e.value.code == -1024
    d.close()


@pytest.mark.gpu
def test_gpu_the_device_entropy_decoder_leaves_pending_verdicts_to_the_host(oracle):
    """The legacy frame of such a file is one sequential scan with restart markers -- what the device entropy decoder takes; the
    verdict that waits behind the frame (HostDecoder::verdict_pending) is the host decoder's to report, in every entropy mode
    (tools/multiscan_campaign.py DAMAGE=1, seed 74: two files read without complaint through "prefer-gpu")."""
    blob = _residual_without_a_specification()
    for mode in ("host", "auto", "prefer-gpu"):
        d = api.Decoder(0)
        with pytest.raises(api.MijpegError) as e:
            d.read(blob, entropy=mode)
        assert e.value.code == -1024, mode
        d.close()
    d = api.Decoder(0)
    with pytest.raises(api.MijpegError) as e:
        d.read(blob, entropy="gpu")
    assert e.value.code == api.ERR_NOT_AVAILABLE
    d.close()


def _rebuild_resi(data, edit):
    """The RESI box's payload (one APP11 segment in this fixture) replaced by edit(payload), lengths adjusted."""
    import struct
    p = data.index(b"RESI") - 16
    assert data[p:p + 2] == b"\xff\xeb"
    ln = struct.unpack(">H", data[p + 2:p + 4])[0]
    seg = data[p:p + 2 + ln]
    payload = edit(seg[20:])
    new = b"\xff\xeb" + struct.pack(">H", 18 + len(payload)) + seg[4:12] + struct.pack(">I", 8 + len(payload)) + seg[16:20] + payload
    return data[:p] + new + data[p + 2 + ln:]


@pytest.mark.parametrize("which", ["no_eoi", "insert_and_no_eoi", "insert_with_eoi", "insert_and_nothing_of_the_eoi"])
def test_a_residual_codestream_that_simply_ends(oracle, which):
    """A residual frame whose data simply ends stands at "its" EOI -- the legacy stream's, which Image::InputStreamOf hands out
    then (codestream/image.cpp:978-996) -- and its hidden refinement scans are read; but only if the scan's decoder stopped at the
    very end of the data: one whose data went wrong (a byte too many) is through with its blocks a byte in front of it, the
    trailer finds garbage, and the RFIN boxes are never looked at (tools/xt_gpu_damage_campaign.py, seed 2001: the planned
    decode applied them).  Residual planes against the oracle's, the oracle's picture against the reference's."""
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "xt_grey", "ghdr_R1_rR3.jpg"), "rb") as f:
        data = f.read()
    edit = {"no_eoi": lambda p: p[:-1], "insert_and_no_eoi": lambda p: p[:700] + b"\x3f" + p[700:-1],
            "insert_with_eoi": lambda p: p[:700] + b"\x3f" + p[700:], "insert_and_nothing_of_the_eoi": lambda p: p[:700] + b"\x3f" + p[700:-2]}[which]
    blob = _rebuild_resi(data, edit)
    codes, _, oerr = oracle.decode_xt_status(blob)
    assert oerr == 0
    if oracle.have_reference():
        assert np.array_equal(oracle.reference_decode_hdr(blob).reshape(-1), oracle.half_codes_to_float(np.asarray(codes)).reshape(-1).astype(np.float32))
    _, rp = oracle.decode_xt_residual_planes(blob)
    refined = bool((np.asarray(rp[0]) % 8 != 0).any())  # three hidden bits: without the RFIN scans every coefficient is a multiple of 8
    assert refined == (which in ("no_eoi", "insert_with_eoi"))
    for threads in (1, 4):
        d = api.Decoder(None)
        d.read(blob, threads)
        assert np.array_equal(d.residual_coefficients(0).reshape(-1).astype(np.int64), np.asarray(rp[0]).reshape(-1).astype(np.int64)), threads
        d.close()
