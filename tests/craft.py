"""Hand-written baseline JPEG streams of any component count (1..4) and any sampling factors (1..4 x 1..4): headers of
our own plus entropy coded data written symbol by symbol over the Annex K tables (taken from a stream Pillow wrote).
Neither Pillow nor the reference's encoder writes two- or four-component frames with mixed sampling, and those are the
frames where the rectangle service's state between calls shows (tests/test_rect_calls.py).  Test infrastructure."""
from __future__ import annotations

import io

import numpy as np

from damage import _BitWriter, _huffman_codes, _segments

_DHT = None


def _annex_k():
    """(DHT segment bytes, code maps) of the four Annex K tables."""
    global _DHT
    if _DHT is None:
        from PIL import Image
        buf = io.BytesIO()
        Image.new("RGB", (16, 16)).save(buf, format="JPEG", quality=75)
        data = buf.getvalue()
        segs = b"".join(data[p:p + 2 + ln] for m, p, ln in _segments(data) if m == 0xC4)
        _DHT = (segs, _huffman_codes(data))
    return _DHT


def layout_blocks(w, h, samp):
    hmax, vmax = max(s[0] for s in samp), max(s[1] for s in samp)
    return (w + 8 * hmax - 1) // (8 * hmax), (h + 8 * vmax - 1) // (8 * vmax)


def craft_stream(rng: np.random.Generator, samp, w: int, h: int, dri: int = 0, tq=None, ac_density: float = 0.08, adobe=None) -> bytes:
    """samp: [(H, V)] per component.  Random smooth-ish content: DC random walk, sparse AC."""
    dht, tabs = _annex_k()
    nc = len(samp)
    tq = tq if tq is not None else [0] + [1] * (nc - 1)
    out = bytearray(b"\xff\xd8")
    if adobe is not None:
        out += b"\xff\xee\x00\x0eAdobe\x00\x64\x00\x00\x00\x00" + bytes([adobe])
    zz = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57,
          50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
    for t in sorted(set(tq)):
        base = rng.integers(2, 14) + (np.arange(64) // 8 + np.arange(64) % 8) * rng.integers(1, 4)
        q = np.clip(base, 1, 255).astype(np.uint8)
        out += b"\xff\xdb\x00\x43" + bytes([t]) + bytes(int(q[zz[k]]) for k in range(64))
    out += b"\xff\xc0" + (8 + 3 * nc).to_bytes(2, "big") + b"\x08" + h.to_bytes(2, "big") + w.to_bytes(2, "big") + bytes([nc])
    for c in range(nc):
        out += bytes([c + 1, (samp[c][0] << 4) | samp[c][1], tq[c]])
    out += dht
    if dri:
        out += b"\xff\xdd\x00\x04" + dri.to_bytes(2, "big")
    out += b"\xff\xda" + (6 + 2 * nc).to_bytes(2, "big") + bytes([nc])
    for c in range(nc):
        t = 0 if c == 0 else 1
        out += bytes([c + 1, (t << 4) | t])
    out += b"\x00\x3f\x00"
    mx, my = layout_blocks(w, h, samp) if nc > 1 else ((w + 7) // 8, (h + 7) // 8)  # one component: the scan is not interleaved
    bw = _BitWriter()

    def put_value(table, run, v):
        s = int(abs(v)).bit_length()
        code, length = table[(run << 4) | s]
        bw.put(code, length)
        if s:
            bw.put(v if v > 0 else v + (1 << s) - 1, s)

    pred = [0] * nc
    level = [int(rng.integers(-40, 40)) for _ in range(nc)]
    rst = 0
    for m in range(mx * my):
        if dri and m and m % dri == 0:
            bw.flush()
            bw.out += bytes([0xFF, 0xD0 + (rst & 7)])
            rst += 1
            pred = [0] * nc
        for c in range(nc):
            t = 0 if c == 0 else 1
            for _b in range(samp[c][0] * samp[c][1] if nc > 1 else 1):
                level[c] = int(np.clip(level[c] + rng.integers(-6, 7), -90, 90))
                put_value(tabs[(0, t)], 0, level[c] - pred[c])
                pred[c] = level[c]
                k = 1
                while k < 64:
                    if rng.random() > ac_density * 6:
                        break
                    run = int(rng.integers(0, 5))
                    if k + run > 63:
                        break
                    v = int(rng.integers(-12, 13)) or 1
                    put_value(tabs[(1, t)], run, v)
                    k += run + 1
                if k <= 63:
                    code, length = tabs[(1, t)][0]
                    bw.put(code, length)
    bw.flush()
    return bytes(out) + bytes(bw.out) + b"\xff\xd9"


def random_layout(rng: np.random.Generator):
    """(samp, w, h, dri): 1..4 components, sampling factors 1..4, at most ten blocks per MCU (the syntax's limit)."""
    while True:
        nc = int(rng.integers(1, 5))
        samp = [(int(rng.integers(1, 5)), int(rng.integers(1, 5))) for _ in range(nc)]
        hmax, vmax = max(s[0] for s in samp), max(s[1] for s in samp)
        # the reference wants subsampling factors that are integers (marker/component.cpp)
        if any(hmax % s[0] or vmax % s[1] for s in samp):
            continue
        if nc > 1 and sum(s[0] * s[1] for s in samp) > 10:
            continue
        return samp, int(rng.integers(1, 141)), int(rng.integers(1, 141)), int(rng.choice([0, 0, 1, 2, 5]))


def to_dnl(data: bytes) -> bytes:
    """The same frame with its number of lines in a DNL marker: SOF Y = 0 and FFDC 0004 <lines> right behind the entropy coded
    data of the first scan (what the reference's encoder writes for -n, marker/frame.cpp:290-300; decoder side
    codestream/entropyparser.cpp:204-249)."""
    d = bytearray(data)
    i, sof, sos = 2, None, None
    while i < len(d):
        assert d[i] == 0xFF
        m, ln = d[i + 1], (d[i + 2] << 8) | d[i + 3]
        if m in (0xC0, 0xC1, 0xC2):
            sof = i
        if m == 0xDA:
            sos = i
            break
        i += 2 + ln
    lines = bytes(d[sof + 5:sof + 7])
    d[sof + 5:sof + 7] = b"\0\0"
    q = sos + 2 + ((d[sos + 2] << 8) | d[sos + 3])
    while not (d[q] == 0xFF and d[q + 1] != 0 and d[q + 1] != 0xFF and not 0xD0 <= d[q + 1] <= 0xD7):
        q += 1
    return bytes(d[:q]) + b"\xff\xdc\x00\x04" + lines + bytes(d[q:])
