#!/usr/bin/env python3
"""Side measurement (not bench.py): encoder direction of the block pipeline, pixels resident in HBM -> coefficient planes in HBM.
8 frames of 8K per launch; algorithmic bytes = pixels in + int16 coefficients out."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from libjpeg_amd import api, synth  # noqa: E402

W, H, F = 7680, 4320, 8
LAY = {"444": ((1, 1, 1), (1, 1, 1)), "420": ((2, 1, 1), (2, 1, 1)), "422": ((2, 1, 1), (1, 1, 1))}
img = synth.synth_image(W, H, 1234)
d = api.Decoder(0)
for sub in os.environ.get("LAYOUTS", "420,422,444").split(","):
    ref = d.read(synth.encode_jpeg(img, 85, sub, restart_mcus=8))  # for its quantiser tables
    info = api.frame_layout(W, H, 3, *LAY[sub], [list(ref.quant[t]) for t in range(4)], quant_index=list(ref.quant_index)[:3])
    px = torch.from_numpy(img).cuda().unsqueeze(0).repeat(F, 1, 1, 1).contiguous()
    n = int(info.coef_count)
    coef = torch.empty((F, n), dtype=torch.int16, device="cuda")
    stream = torch.cuda.current_stream()

    def step():
        api.launch_forward(info, px.data_ptr(), coef.data_ptr(), F, W * 3, H * W * 3, stream=stream.cuda_stream)

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(10):
        step()
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    bpp = 3 + 2.0 * n / (W * H)
    # how close to the picture Pillow's encoder saw: decode our coefficients with the fused kernel
    out = torch.empty((1, H, W * 3), dtype=torch.uint8, device="cuda")
    api.launch_reconstruct(ref, coef.data_ptr(), out.data_ptr(), 1, W * 3, H * W * 3, n, stream=stream.cuda_stream)
    torch.cuda.synchronize()
    err = (out[0].cpu().numpy().reshape(H, W, 3).astype(np.int16) - img).astype(np.float64)
    print(f"{sub}: {'fdct420_tile_kernel' if sub == '420' else 'fdct_interior_kernel x3'} + fdct_blocks_kernel {ms:7.3f} ms/launch {W*H*F/ms/1e6:8.1f} Gpixel/s {W*H*F*bpp/ms/1e6:7.0f} GB/s algorithmic ({bpp:.1f} B/px); "
          f"round trip through the fused decoder: PSNR {10*np.log10(255**2/np.mean(err**2)):.1f} dB", flush=True)
# whole pipeline for one 8K picture in host memory (upload, kernels, download of the coefficients, entropy coder on the host cores)
import time
for sub, ri, opt, coder in (("420", 8, False, "gpu"), ("420", 8, True, "gpu"), ("420", 0, False, "gpu"), ("444", 0, True, "gpu"),
                            ("420", 8, False, "host"), ("420", 8, True, "host"), ("420", 0, False, "host")):
    ts = []
    d.encode(img, 85, sub, ri, opt, coder=coder)  # warm: pinned staging, worker threads
    for _ in range(5):
        t = time.perf_counter(); data = d.encode(img, 85, sub, ri, opt, coder=coder); ts.append(time.perf_counter() - t)
    print(f"encode one 8K {sub} picture, entropy coder on the {'device' if coder == 'gpu' else 'host'}, restart interval {ri}, {'optimised' if opt else 'Annex K'} Huffman tables: {min(ts)*1e3:.1f} ms "
          f"= {W*H/min(ts)/1e6:.0f} Mpixel/s, {len(data)/1e6:.2f} MB; gather+upload / kernels / download / coder ms: "
          f"{[round(v * 1e3, 2) for v in d.timing().values()]}", flush=True)
# frames that are already in HBM: forward kernels for the batch, device coder frame by frame, only streams come down
import ctypes as C
luma, chroma = np.zeros(64, np.uint16), np.zeros(64, np.uint16)
L = api.lib()
L.mijpeg_quality_tables.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
L.mijpeg_quality_tables.restype = None
L.mijpeg_quality_tables(85, luma.ctypes.data, chroma.ctypes.data)
binfo = api.frame_layout(W, H, 3, (2, 1, 1), (2, 1, 1), [luma, chroma], quant_index=[0, 0, 0])
NB = 16
bpx = torch.from_numpy(img).cuda().unsqueeze(0).repeat(NB, 1, 1, 1).contiguous()
bcoef = torch.empty((NB, int(binfo.coef_count)), dtype=torch.int16, device="cuda")
d.encode_batch_device(binfo, bpx.data_ptr(), bcoef.data_ptr(), 2, W * 3, H * W * 3, 8, False)
ts = []
for _ in range(3):
    ss = d.encode_batch_device(binfo, bpx.data_ptr(), bcoef.data_ptr(), NB, W * 3, H * W * 3, 8, False)
    ts.append(list(d.timing().values())[0])  # the C call alone (the Python wrapper copies the streams once more)
print(f"encode {NB} 8K 420 frames resident in HBM (restart interval 8, Annex K tables): {min(ts)*1e3:.1f} ms = {min(ts)*1e3/NB:.2f} ms per frame, "
      f"{W*H*NB/min(ts)/1e6:.0f} Mpixel/s, {len(ss[0])/1e6:.2f} MB each", flush=True)
del bpx, bcoef
d.close()
# the reference encoder on one host core, same picture and switches
from oracle import oracle as O
if O.have_reference():
    t = time.perf_counter(); ref = O.reference_encode(img, ["-bl", "-q", "85", "-s", "1x1,2x2,2x2", "-z", "8"]); dt = time.perf_counter() - t
    print(f"reference encoder (oracle/_ref/jpeg, one core, file in and out of /dev/shm): {dt*1e3:.0f} ms = {W*H/dt/1e6:.1f} Mpixel/s, {len(ref)/1e6:.2f} MB")
