#!/usr/bin/env python3
"""Side measurement (not bench.py): JPEG XT profile C, 4K, one GPU -- BASELINE config 5.
The HDR stream is produced with the reference encoder (oracle/_ref/jpeg), so this is a tools/ script."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from libjpeg_amd import api, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

W, H, F = 3840, 2160, 8
data = O.reference_encode_hdr(synth.synth_hdr(W, H, 99), ["-r", "-q", "85", "-Q", "90", "-h", "-profile", "c", "-r12", "-s", "1x1,2x2,2x2"])
d = api.Decoder(0)
t = time.perf_counter(); info = d.read(data); t_read = time.perf_counter() - t
reads = {}
for mode in ("host", "gpu"):
    ts = []
    for _ in range(4):
        t = time.perf_counter(); d.read(data, entropy=mode); ts.append(time.perf_counter() - t)
    reads[mode] = min(ts) * 1e3
info = d.read(data)
xt = d.xt_params()
n = int(info.coef_count)
host = np.concatenate([d.coefficients(c).reshape(-1) for c in range(3)])
# residual planes follow the legacy planes in the decoder's buffer: fetch the whole frame from the device copy
coef = torch.empty((F, n), dtype=torch.int16, device="cuda")
import ctypes as C
src = d.device_coefficients()
torch.cuda.synchronize()
for f in range(F):
    assert api.lib() and torch.cuda.current_stream()
    C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(coef[f].data_ptr()), C.c_void_p(src), C.c_size_t(n * 2), 3)
row = W * 3 * 2
out = torch.empty((F, H, row), dtype=torch.uint8, device="cuda")
wsb = api.workspace_bytes(info, F)
ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream()
def step():
    api.launch_reconstruct(info, coef.data_ptr(), out.data_ptr(), F, row, H * row, n, workspace=ws.data_ptr(), workspace_bytes=wsb,
                           stream=stream.cuda_stream, xt=xt)
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(stream)
for _ in range(10): step()
e1.record(stream); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
# algorithmic bytes (SURVEY 8d, config 5 with a 4:2:0 base): base 3 B/px + residual 4:4:4 int16 6 B/px + out half 6 B/px
bpp = 3 + 6 + 6
print(f"XT profile C {W}x{H} x{F}: {ms:.3f} ms/launch = {W*H*F/ms/1e3:.0f} Mpix/s, {W*H*F*bpp/ms/1e6:.0f} GB/s algorithmic ({bpp} B/px); "
      f"read (Huffman x2 + upload) {t_read*1e3:.1f} ms for one frame, {len(data)/1e6:.1f} MB stream")
exp, _ = O.decode_xt(data)
got = out[0].cpu().numpy().view(np.uint16).reshape(H, W, 3)
print("frame 0 bit-exact vs oracle:", bool(np.array_equal(got, exp)))
print(f"read of one frame (both codestreams to coefficients in HBM): host entropy decoder {reads['host']:.2f} ms, device entropy decoder {reads['gpu']:.2f} ms")
