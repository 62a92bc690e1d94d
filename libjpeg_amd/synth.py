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
