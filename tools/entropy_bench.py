"""Host vs on-device entropy decoding of one 8K 4:2:0 frame (wall clock of Decoder.read, coefficients end up in HBM)."""
import sys
import time

import numpy as np

import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from libjpeg_amd import api, synth

W, H = 7680, 4320
img = synth.synth_image(W, H, 1234)
import os
for ri in [int(x) for x in os.environ.get('RI', '1,2,4,8,32').split(',')]:
    data = synth.encode_jpeg(img, 85, "420", restart_mcus=ri)
    d = api.Decoder(0)
    res = {}
    for mode in ("host", "gpu") if ri else ("host",):
        ts = []
        for it in range(6):
            t0 = time.perf_counter()
            d.read(data, entropy=mode)
            ts.append(time.perf_counter() - t0)
        res[mode] = min(ts) * 1e3
        res[mode + "_t"] = {k: round(v * 1e3, 3) for k, v in d.timing().items()}
        t0 = time.perf_counter()
        out = d.reconstruct()
        res[mode + "_rec"] = (time.perf_counter() - t0) * 1e3
        res[mode + "_sum"] = int(out.astype(np.uint64).sum())
    if not ri:
        print(f"dri={ri:4d} bytes={len(data)/1e6:.2f}MB host read {res['host']:.2f} ms {res['host_t']}", flush=True)
        continue
    assert res["host_sum"] == res["gpu_sum"]
    print(f"dri={ri:4d} bytes={len(data)/1e6:.2f}MB host read {res['host']:.2f} ms  gpu read {res['gpu']:.2f} ms  {res['gpu_t']}", flush=True)
    d.close()


# streams without restart markers: sequential (1 thread) vs the self-synchronising parallel host decoder
data = synth.encode_jpeg(img, 85, "420", restart_mcus=0)
for th in (1, 4, 16, 64):
    d = api.Decoder(None)
    ts = []
    for it in range(3):
        t0 = time.perf_counter()
        d.read(data, threads=th)
        ts.append(time.perf_counter() - t0)
    print(f"threads_sweep no-DRI 8K: threads={th:3d} read {min(ts)*1e3:.2f} ms  speculative scans/pieces so far {api.speculative_scans()}", flush=True)
    d.close()


d = api.Decoder(0)
for mode, walk in (("host", "1"), ("gpu", "0"), ("gpu", "1")):
    os.environ["MIJPEG_DEVICE_WALK"] = walk
    ts = []
    for it in range(6):
        t0 = time.perf_counter()
        d.read(data, entropy=mode)
        ts.append(time.perf_counter() - t0)
    where = "" if mode == "host" else (" (walk on the device, %d rounds)" % d.device_walk_rounds() if walk == "1" else " (walk on the host)")
    print(f"no-DRI 8K to coefficients in HBM, entropy={mode}{where}: {min(ts)*1e3:.2f} ms  { {k: round(v * 1e3, 3) for k, v in d.timing().items()} }", flush=True)
os.environ["MIJPEG_DEVICE_WALK"] = "1"
d.close()

# batches of frames on the device: one Huffman launch + one reconstruction launch for n frames, pixels left in HBM
import torch
for dri, walk, sizes in ((8, "1", (1, 4, 16, 32)), (0, "1", (1, 4, 16)), (0, "0", (4, 16))):
    if os.environ.get("BATCH_ONLY_NODRI") and dri:
        continue
    os.environ["MIJPEG_DEVICE_WALK"] = walk
    frames = [synth.encode_jpeg(synth.synth_image(W, H, 2000 + i), 85, "420", restart_mcus=dri) for i in range(4)]
    what = f"DRI {dri}" if dri else ("no DRI, walk on the device" if walk == "1" else "no DRI, walk on the host")
    for n in sizes:
        batch = [frames[i % 4] for i in range(n)]
        d = api.Decoder(0)
        out = torch.empty((n, H, W * 3), dtype=torch.uint8, device="cuda")
        ts, tp = [], []
        for it in range(4):
            t0 = time.perf_counter()
            d.decode_batch_device(batch)
            t1 = time.perf_counter()
            d.reconstruct_batch_device(out.data_ptr(), H * W * 3, W * 3)
            ts.append(time.perf_counter() - t0)
            tm = d.timing()
            tp.append((round(tm["h2d_wait"] * 1e3, 2), round(tm["kernel"] * 1e3, 2), round(tm["d2h"] * 1e3, 2), round((time.perf_counter() - t1) * 1e3, 2)))
        rounds = f", {d.device_walk_rounds()} rounds" if d.device_walk_rounds() else ""
        print(f"batch of {n:2d} 8K frames ({what}{rounds}): {min(ts)*1e3:7.2f} ms = {min(ts)*1e3/n:.3f} ms per frame, {W*H*n/min(ts)/1e6:8.0f} Mpixel/s (parse, prepare, upload+huffman, reconstruct ms: {tp[-1]})", flush=True)
        d.close()
os.environ["MIJPEG_DEVICE_WALK"] = "1"
