"""Seeded stream corruptions for the damaged-stream parity tests (test infrastructure).

What the reference does with damaged streams is restated in oracle/jpeg_oracle.c (rs_* functions:
codestream/entropyparser.cpp:117-201 resynchronisation, io/bitstream.cpp:56-137 bit reader at markers / at the end of
the data, the warn-and-continue paths of marker/frame.cpp and codestream/tables.cpp) and pinned against oracle/_ref/jpeg
by tests/test_damaged.py; the product (libjpeg_amd) is compared with the oracle on the same corruptions.
"""
from __future__ import annotations

import numpy as np

KINDS = ("flip1", "flip3", "flipbits", "drop_interval", "dup_interval", "swap_intervals", "truncate", "insert",
         "zero_run", "ff_run", "renumber_rst", "drop_marker_only")


def entropy_start(data: bytes) -> int:
    """Offset of the first byte behind the first SOS header (0 if there is none)."""
    p = 2
    n = len(data)
    while p + 4 <= n:
        if data[p] != 0xFF:
            p += 1
            continue
        m = data[p + 1]
        if m == 0xFF:
            p += 1
            continue
        if m in (0x01,) or 0xD0 <= m <= 0xD9:
            p += 2
            continue
        ln = (data[p + 2] << 8) | data[p + 3]
        if m == 0xDA:
            return min(n, p + 2 + ln)
        p += 2 + ln
    return 0


def restart_markers(data: bytes, start: int = 0):
    """Offsets of all FF D0..D7 pairs from `start` on."""
    a = np.frombuffer(data, np.uint8)
    idx = np.nonzero((a[:-1] == 0xFF) & (a[1:] >= 0xD0) & (a[1:] <= 0xD7))[0]
    return [int(i) for i in idx if i >= start]


def corrupt(data: bytes, kind: str, rng: np.random.Generator, where: str = "any") -> bytes:
    """One seeded corruption.  where: "any" (whole file behind SOI), "entropy" (behind the first scan header) or
    "header" (in front of it)."""
    d = bytearray(data)
    n = len(d)
    es = entropy_start(data)
    lo, hi = 2, n
    if where == "entropy" and es:
        lo = es
    elif where == "header" and es:
        hi = es
    if hi <= lo:
        lo, hi = 2, n

    def pos():
        return int(rng.integers(lo, hi))

    if kind == "flip1":
        d[pos()] = int(rng.integers(0, 256))
    elif kind == "flip3":
        for _ in range(3):
            d[pos()] = int(rng.integers(0, 256))
    elif kind == "flipbits":
        for _ in range(int(rng.integers(1, 4))):
            d[pos()] ^= 1 << int(rng.integers(0, 8))
    elif kind in ("drop_interval", "dup_interval", "swap_intervals", "renumber_rst", "drop_marker_only"):
        ms = restart_markers(data, es)
        if len(ms) < 3:
            return corrupt(data, "flip3", rng, "entropy")
        i = int(rng.integers(0, len(ms) - 2))
        a, b, c = ms[i], ms[i + 1], ms[i + 2]
        if kind == "drop_interval":  # marker i and the data behind it
            del d[a:b]
        elif kind == "dup_interval":
            d[b:b] = d[a:b]
        elif kind == "swap_intervals":
            d[a:c] = d[b:c] + d[a:b]
        elif kind == "renumber_rst":
            d[a + 1] = 0xD0 + int(rng.integers(0, 8))
        else:
            del d[a:a + 2]
    elif kind == "truncate":
        del d[pos():]
    elif kind == "insert":
        p = pos()
        d[p:p] = bytes(int(x) for x in rng.integers(0, 256, int(rng.integers(1, 9))))
    elif kind == "zero_run":
        p = pos()
        ln = int(rng.integers(1, 33))
        d[p:p + ln] = bytes(min(ln, n - p))
    elif kind == "ff_run":
        p = pos()
        ln = int(rng.integers(1, 5))
        d[p:p + ln] = b"\xff" * min(ln, n - p)
    else:
        raise ValueError(kind)
    return bytes(d)


def cases(data: bytes, count: int, seed: int, where: str = "any"):
    """`count` seeded (kind, corrupted stream) pairs, cycling through KINDS."""
    rng = np.random.default_rng(seed)
    for i in range(count):
        kind = KINDS[i % len(KINDS)]
        yield kind, corrupt(data, kind, rng, where)
