#!/usr/bin/env python3
"""tests/golden/xt_boxes/: a curve whose table leaves 32 bits (tests/test_xt_boxes.py, "tables beyond the range").
xt_int8/a_r2_exponential.jpg with the second parameter of its CURV box at 1024.0 (first byte of the IEEE number: 0x3f -> 0x44);
the .bin holds the samples oracle/_ref/jpeg (the reference decoder, `make -C oracle ref`) writes for it.  Build container only.
refused_spec_damaged_residual.jpg is not made here: it is stream `a_q_is_tone_box / hdr8` of `python tools/box_campaign.py xt 9
--keep DIR` (xt_int8/a_q_is_tone_box.jpg with bytes 556 and 1426 changed to e0 / 2d), kept as the campaign wrote it."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def main():
    with open(os.path.join(HERE, "xt_int8", "a_r2_exponential.jpg"), "rb") as f:
        data = bytearray(f.read())
    at = data.index(b"CURV") + 4 + 2 + 4  # payload: index/type, rounding, P1 (4 bytes), then P2
    assert data[at:at + 4] == b"\x3f\x80\x00\x00"  # 1.0
    data[at] = 0x44
    out = os.path.join(HERE, "xt_boxes")
    os.makedirs(out, exist_ok=True)
    jpg = os.path.join(out, "r2_curve_overflows.jpg")
    with open(jpg, "wb") as f:
        f.write(bytes(data))
    ppm = os.path.join(out, "tmp.ppm")
    subprocess.run([O.REF_BIN, jpg, ppm], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    O.read_pnm_any(ppm).astype("uint8").tofile(os.path.join(out, "r2_curve_overflows.bin"))
    os.remove(ppm)


if __name__ == "__main__":
    main()
