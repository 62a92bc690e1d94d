"""Deterministic synthetic test images and JPEG streams (SURVEY.md §8d recipe).

img[y,x,c] = clip(128 + A_c * f_c(x,y) + N(0, 6), 0, 255) with
f_0 = sin(x/37) cos(y/53), f_1 = sin((x+y)/91), f_2 = cos(x/17 - y/29), A = (100, 90, 80).

The JPEG streams are produced with Pillow (libjpeg-turbo) -- a third-party encoder that is part of
this image, not the reference and not the oracle: it is only a source of *valid baseline bitstreams*
with the sampling / restart layout a test or bench asks for.
"""
from __future__ import annotations

import io

import numpy as np

_SUBSAMPLING = {"444": 0, "422": 1, "420": 2}


def synth_image(width: int, height: int, seed: int = 1234, channels: int = 3) -> np.ndarray:
    rng = np.random.default_rng(seed)
    x = np.arange(width, dtype=np.float32)[None, :]
    y = np.arange(height, dtype=np.float32)[:, None]
    f = [np.sin(x / 37.0) * np.cos(y / 53.0), np.sin((x + y) / 91.0), np.cos(x / 17.0 - y / 29.0)]
    amp = (100.0, 90.0, 80.0)
    out = np.empty((height, width, channels), dtype=np.uint8)
    for c in range(channels):
        noise = rng.normal(0.0, 6.0, size=(height, width)).astype(np.float32)
        out[..., c] = np.clip(128.0 + amp[c % 3] * f[c % 3] + noise, 0, 255).astype(np.uint8)
    return out


def encode_jpeg(img: np.ndarray, quality: int = 85, subsampling: str = "420",
                restart_mcus: int = 0, optimize: bool = False, progressive: bool = False) -> bytes:
    """Baseline (SOF0) Huffman JPEG via Pillow; restart_mcus = DRI value (0 = none)."""
    from PIL import Image

    if img.ndim == 3 and img.shape[2] == 1:
        img = img[..., 0]
    im = Image.fromarray(img, "L" if img.ndim == 2 else "RGB")
    buf = io.BytesIO()
    kw = dict(format="JPEG", quality=quality, optimize=optimize)
    if img.ndim == 3:
        kw["subsampling"] = _SUBSAMPLING[subsampling]
    if restart_mcus:
        kw["restart_marker_blocks"] = restart_mcus
    if progressive:
        kw["progressive"] = True
    im.save(buf, **kw)
    return buf.getvalue()


def synth_jpeg(width: int, height: int, seed: int = 1234, quality: int = 85,
               subsampling: str = "420", restart_mcus: int = 0, progressive: bool = False) -> bytes:
    return encode_jpeg(synth_image(width, height, seed), quality, subsampling, restart_mcus, progressive=progressive)


def synth_hdr(width: int, height: int, seed: int = 99) -> np.ndarray:
    """Synthetic HDR picture: base^2.2 * 2^(4 sin(x/400) + 2 cos(y/300)) * (1 + N(0, 0.01)),
    floor 1e-4, float32 RGB."""
    rng = np.random.default_rng(seed)
    base = synth_image(width, height, seed).astype(np.float32) / 255.0
    y, x = np.mgrid[0:height, 0:width].astype(np.float32)
    gain = np.exp2(4.0 * np.sin(x / 400.0) + 2.0 * np.cos(y / 300.0))[..., None]
    noise = 1.0 + rng.normal(0.0, 0.01, size=(height, width, 3)).astype(np.float32)
    return np.maximum(base ** 2.2 * gain * noise, 1e-4).astype(np.float32)


def to_12bit(jpeg8: bytes, scale: int = 16) -> bytes:
    """An 8-bit Huffman sequential JPEG (baseline tables, 8-bit quantiser entries) -> the 12-bit extended sequential stream
    (SOF1, P = 12) with the same entropy coded data and every quantiser delta multiplied by `scale` (16-bit DQT entries): the
    picture it decodes to is the 8-bit one times `scale` -- synthetic 12-bit content of any size without a 12-bit encoder.
    (There is no network and no 12-bit Pillow; the reference's own encoder writes the small 12-bit fixtures of tests/golden.)"""
    out = bytearray(jpeg8[:2])
    p = 2
    while p + 4 <= len(jpeg8):
        assert jpeg8[p] == 0xFF, "marker expected"
        m = jpeg8[p + 1]
        ln = (jpeg8[p + 2] << 8) | jpeg8[p + 3]
        seg = jpeg8[p + 4:p + 2 + ln]
        if m == 0xDB:  # DQT: Pq = 0 tables -> Pq = 1, deltas * scale
            q, body = 0, bytearray()
            while q < len(seg):
                pq, tq = seg[q] >> 4, seg[q] & 15
                assert pq == 0, "8-bit quantiser tables expected"
                body.append(0x10 | tq)
                for v in seg[q + 1:q + 65]:
                    w = min(65535, v * scale)
                    body += bytes((w >> 8, w & 255))
                q += 65
            out += bytes((0xFF, 0xDB, (len(body) + 2) >> 8, (len(body) + 2) & 255)) + body
        elif m == 0xC0 or m == 0xC1:  # SOF0 / SOF1, P = 8 -> SOF1, P = 12
            out += bytes((0xFF, 0xC1)) + jpeg8[p + 2:p + 4] + bytes((12,)) + seg[1:]
        elif m == 0xC2:  # a progressive frame stays one: SOF2, P = 12 (its later scans carry DHT segments of their own: copied as they are)
            out += bytes((0xFF, 0xC2)) + jpeg8[p + 2:p + 4] + bytes((12,)) + seg[1:]
        else:
            out += jpeg8[p:p + 2 + ln]
        p += 2 + ln
        if m == 0xDA:
            out += jpeg8[p:]  # (everything behind the first scan header as it is: later scans' tables are Huffman tables, not DQT)
            break
    return bytes(out)
