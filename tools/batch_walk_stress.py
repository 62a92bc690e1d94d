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
# two sets of streams, alternating: what an earlier decode left in the buffers is never the data of the current one
sets = [[synth.encode_jpeg(synth.synth_image(w, h, base + i), 85, "420", restart_mcus=0, optimize=(i % 3 == 1)) for i in range(n)] for base in (500, 900)]
one = api.Decoder(0)
refs = []
for streams in sets:
    ref = []
    for st in streams:
        one.read(st)  # host entropy decoder
        ref.append(one.reconstruct().copy())
    refs.append(ref)
row = w * 3
bad = 0
shared = api.Decoder(0)  # one object across all repeats as well: its buffers are reused with the other set's leftovers
for r in range(R):
    streams, ref = sets[r & 1], refs[r & 1]
    d = api.Decoder(0) if r % 3 else shared
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
    if d is not shared:
        d.close()
print("differences:", bad, "of", 2 * R * n, "frame decodes")
