#!/usr/bin/env python3
"""Stress of the batched device walk (streams without restart markers, pipelined uploads): decode the same batch R times and
compare every frame with a single-image decode of the same stream; prints where a frame differs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from libjpeg_amd import api, synth  # noqa: E402

w, h, n = 3840, 2160, 10
R = int(os.environ.get("REPEATS", "12"))
streams = [synth.encode_jpeg(synth.synth_image(w, h, 500 + i), 85, "420", restart_mcus=0, optimize=(i % 3 == 1)) for i in range(n)]
one = api.Decoder(0)
ref = []
for st in streams:
    one.read(st)  # host entropy decoder
    ref.append(one.reconstruct().copy())
row = w * 3
bad = 0
for r in range(R):
    d = api.Decoder(0)
    d.decode_batch_device(streams, min_intervals=1)
    rounds = d.device_walk_rounds()
    out = torch.zeros((n, h, row), dtype=torch.uint8, device="cuda")
    d.reconstruct_batch_device(out.data_ptr(), h * row, row)
    res = out.cpu().numpy().reshape(n, h, w, 3)
    for i in range(n):
        diff = np.argwhere((res[i] != ref[i]).any(axis=2))
        if diff.size:
            bad += 1
            print(f"repeat {r} frame {i}: {len(diff)} pixels differ, rows {diff[:,0].min()}..{diff[:,0].max()} cols {diff[:,1].min()}..{diff[:,1].max()} rounds {rounds}", flush=True)
    # again on the same object: buffers are reused
    d.decode_batch_device(streams, min_intervals=1)
    d.reconstruct_batch_device(out.data_ptr(), h * row, row)
    res = out.cpu().numpy().reshape(n, h, w, 3)
    for i in range(n):
        if not np.array_equal(res[i], ref[i]):
            bad += 1
            print(f"repeat {r} (second call) frame {i} differs", flush=True)
    d.close()
print("differences:", bad, "of", 2 * R * n, "frame decodes")
