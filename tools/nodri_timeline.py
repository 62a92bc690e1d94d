#!/usr/bin/env python3
"""One 8K 4:2:0 stream WITHOUT restart markers, bytes in host memory -> pixels in HBM, entropy decoding on the device
(huffman_walk_kernel rounds + huffman_scan_kernel).  Prints wall times; under `rocprofv3 --kernel-trace --memory-copy-trace`
tools/nodri_timeline.sh turns the trace of the last decode into a timeline."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from libjpeg_amd import api, synth  # noqa: E402

W, H = int(os.environ.get("W", "7680")), int(os.environ.get("H", "4320"))
N = int(os.environ.get("FRAMES", "1"))
data = [synth.synth_jpeg(W, H, seed=1234 + i, quality=85, subsampling="420", restart_mcus=0) for i in range(min(N, 2))]
row = W * 3
dec = api.Decoder(0)
if N == 1:
    out = torch.empty((H, row), dtype=torch.uint8, device="cuda")
    ts = []
    for it in range(8):
        torch.cuda.synchronize()
        t = time.perf_counter()
        dec.read(data[0], entropy="gpu")
        dec.reconstruct_device(out.data_ptr(), row)
        ts.append(time.perf_counter() - t)
    print("single frame ms:", " ".join("%.2f" % (x * 1e3) for x in ts), "rounds", dec.device_walk_rounds(), "stream bytes", len(data[0]))
else:
    out = torch.empty((N, H, row), dtype=torch.uint8, device="cuda")
    ts = []
    for it in range(5):
        torch.cuda.synchronize()
        t = time.perf_counter()
        dec.decode_batch_device([data[i % 2] for i in range(N)])
        dec.reconstruct_batch_device(out.data_ptr(), H * row, row)
        ts.append(time.perf_counter() - t)
    print("batch of %d ms/frame:" % N, " ".join("%.3f" % (x * 1e3 / N) for x in ts))
