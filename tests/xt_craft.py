"""Hand-made JPEG XT profile C variants (test infrastructure): streams the reference ENCODER's command line cannot write but
its DECODER accepts -- parametric curves (CURV boxes) as L, Q and R2 tables, a residual DCT bypass (RDCT box) -- made by
editing the merging specification of a stream the reference encoder wrote.  Box syntax: APP11 segments `FF EB len 'JP' En(2)
Z(4) LBox(4) TBox(4) payload` (boxes/box.cpp:88-200), the SPEC box is a superbox of LBox/TBox/payload children
(boxes/superbox.cpp); CURV payload boxes/parametrictonemappingbox.cpp:84-149; index boxes boxes/nonlineartrafobox.cpp:62-80."""
from __future__ import annotations

import struct

CURVE = dict(zero=0, constant=1, identity=2, gamma=4, linear=5, exponential=6, logarithmic=7, gammaoffset=8)


def _segments(data: bytes):
    """(offset, marker, length incl. the two length bytes) of the marker segments in front of the first SOS."""
    p = 2
    while p + 4 <= len(data):
        m = data[p + 1]
        ln = (data[p + 2] << 8) | data[p + 3]
        yield p, m, ln
        if m == 0xDA:
            return
        p += 2 + ln


def app11(tbox: bytes, payload: bytes, en: int = 1) -> bytes:
    body = b"JP" + struct.pack(">HIL", en, 1, 8 + len(payload)) + tbox + payload
    return b"\xff\xeb" + struct.pack(">H", 2 + len(body)) + body


def subbox(tbox: bytes, payload: bytes) -> bytes:
    return struct.pack(">L", 8 + len(payload)) + tbox + payload


def curv(index: int, kind: str, e: int = 0, p=(0.0, 0.0, 0.0, 0.0)) -> bytes:
    return bytes([(index << 4) | CURVE[kind], e << 4]) + b"".join(struct.pack(">f", x) for x in p)


def edit_spec(data: bytes, add_children: bytes = b"", drop=(), new_boxes: bytes = b"") -> bytes:
    """Append children to the SPEC superbox (dropping children whose type is in `drop`) and put `new_boxes` (whole APP11
    segments) in front of it."""
    for off, m, ln in _segments(data):
        seg = data[off:off + 2 + ln]
        if m == 0xEB and seg[4:6] == b"JP" and seg[16:20] == b"SPEC":
            payload = seg[20:]
            kept = b""
            j = 0
            while j + 8 <= len(payload):
                l = struct.unpack(">L", payload[j:j + 4])[0]
                if payload[j + 4:j + 8] not in drop:
                    kept += payload[j:j + l]
                j += l
            en = struct.unpack(">H", seg[6:8])[0]
            return data[:off] + new_boxes + app11(b"SPEC", kept + add_children, en) + data[off + 2 + ln:]
    raise ValueError("no SPEC box")


def variants(data: bytes) -> dict[str, bytes]:
    """name -> stream.  Table indices 8..15 are free in what the encoder writes (it uses 0..2 for its TONE boxes)."""
    v = {}
    pts = lambda t, i: subbox(t, bytes([(i << 4) | i, (i << 4) | i]))  # noqa: E731 -- the same table for all components
    v["q_linear"] = edit_spec(data, pts(b"QPTS", 8), new_boxes=app11(b"CURV", curv(8, "linear", 0, (0.1, 0.9, 0, 0))))
    v["r2_linear"] = edit_spec(data, pts(b"RPTS", 9), new_boxes=app11(b"CURV", curv(9, "linear", 0, (-0.5, 1.5, 0, 0))))
    v["q_and_r2"] = edit_spec(data, pts(b"QPTS", 8) + pts(b"RPTS", 9),
                              new_boxes=app11(b"CURV", curv(8, "linear", 0, (0.05, 0.95, 0, 0))) + app11(b"CURV", curv(9, "linear", 0, (-0.25, 1.25, 0, 0)), 2))
    v["r2_gamma"] = edit_spec(data, pts(b"RPTS", 10), new_boxes=app11(b"CURV", curv(10, "gamma", 0, (0.04045, 2.4, 0.055, 0))))
    v["r2_exponential"] = edit_spec(data, pts(b"RPTS", 11), new_boxes=app11(b"CURV", curv(11, "exponential", 0, (0.0, 1.0, 0.5, -0.4))))
    v["r2_logarithmic"] = edit_spec(data, pts(b"RPTS", 12), new_boxes=app11(b"CURV", curv(12, "logarithmic", 0, (4.0, 1.0, 1.0, -0.3))))
    v["r2_gammaoffset"] = edit_spec(data, pts(b"RPTS", 13), new_boxes=app11(b"CURV", curv(13, "gammaoffset", 0, (0.1, 0.9, 0.8, 0))))
    v["q_identity_curve"] = edit_spec(data, pts(b"QPTS", 8), new_boxes=app11(b"CURV", curv(8, "identity", 0)))
    v["q_constant"] = edit_spec(data, pts(b"QPTS", 8), new_boxes=app11(b"CURV", curv(8, "constant", 0)))
    v["r2_zero"] = edit_spec(data, pts(b"RPTS", 8), new_boxes=app11(b"CURV", curv(8, "zero", 0)))
    v["l_gamma_curve"] = edit_spec(data, pts(b"LPTS", 14), drop=(b"LPTS",), new_boxes=app11(b"CURV", curv(14, "gamma", 1, (0.04045, 2.4, 0.055, 0))))
    v["l_curve_inside_spec"] = edit_spec(data, pts(b"LPTS", 14) + subbox(b"CURV", curv(14, "linear", 1, (0.0, 1.0, 0, 0))), drop=(b"LPTS",))
    v["bypass"] = edit_spec(data, subbox(b"RDCT", b"\x30"))
    v["bypass_noise"] = edit_spec(data, subbox(b"RDCT", b"\x31"))
    v["q_table_missing"] = edit_spec(data, pts(b"QPTS", 15))
    v["q_is_tone_box"] = edit_spec(data, pts(b"QPTS", 0))  # an explicit table where fractional bits are needed: INVALID_PARAMETER
    # R = identity WITHOUT clamping and without the lossless flag, half float output: the Q table is indexed with the samples as they
    # are (colortrafo/ycbcrtrafo.cpp:797-801) -- the default table must exist as a real table there
    v["r_identity_noclamp"] = edit_spec(data, subbox(b"RTRF", b"\x10") + subbox(b"OCON", b"\x84\x00\x00"), drop=(b"RTRF", b"OCON", b"RDCT"))
    v["r2_linear_negative_slope"] = edit_spec(data, pts(b"RPTS", 9), new_boxes=app11(b"CURV", curv(9, "linear", 0, (0.9, 0.1, 0, 0))))
    return v
