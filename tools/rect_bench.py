#!/usr/bin/env python3
"""The drop-in decode of one 8K frame (host entropy decoder, kernel, device-to-host copy, copy into the caller's bitmap):
whole frame in one request, and the stripe loop of cmd/reconstruct.cpp (8 lines per request) with the time to the first stripe."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401

from libjpeg_amd import api, synth  # noqa: E402


def main():
    W, H = 7680, 4320
    data = synth.synth_jpeg(W, H, 1234, 85, "420", 8)
    dec = api.Decoder(0)
    user = np.zeros((H, W, 3), np.uint8)
    for mode in ("whole", "stripes"):
        ts, first = [], []
        for _ in range(6):
            t = time.perf_counter()
            dec.read(data)
            t1 = time.perf_counter()
            if mode == "whole":
                dec.reconstruct(out=user)
                tf = time.perf_counter()
            else:
                tf = None
                for y in range(0, H, 8):
                    dec.reconstruct_rect(0, y, W - 1, min(H, y + 8) - 1, out=user)
                    if tf is None:
                        tf = time.perf_counter()
            ts.append(time.perf_counter() - t)
            first.append(tf - t1)
        print(f"{mode:8s} band {os.environ.get('MIJPEG_RECT_BAND_MIB', 'default'):8s}: total {min(ts) * 1e3:6.2f} ms (median {sorted(ts)[3] * 1e3:6.2f}), "
              f"read -> {'frame' if mode == 'whole' else 'first stripe'} {min(first) * 1e3:6.2f} ms")
    dec.close()


if __name__ == "__main__":
    main()
