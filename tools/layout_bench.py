#!/usr/bin/env python3
"""Side measurement (not bench.py): reconstruction throughput per sampling layout, coefficients resident in HBM.
8 frames of 8K per launch; prints kernel(s), ms per launch, Gpixel/s and algorithmic GB/s."""
import os
import sys
import ctypes as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from libjpeg_amd import api, synth  # noqa: E402

W, H, F = int(os.environ.get("W", "7680")), int(os.environ.get("H", "4320")), 8  # W=7678: lines that start at any byte address
hip = C.cdll.LoadLibrary("libamdhip64.so")
HEADER_EDITS = {"3x1": ((3, 1, 1), (1, 1, 1)), "1x4": ((1, 1, 1), (4, 1, 1)), "lumasub": ((1, 2, 2), (1, 2, 2)), "3x3": ((3, 1, 1), (3, 1, 1))}


def run_layout(label, info, coef, d):
    """20 timed launches of F frames; d: decoder object that holds frame 0's stream (its own reconstruction is compared) or None."""
    n = int(info.coef_count)
    nc = info.components
    p12 = info.precision > 8
    sb = 2 if p12 else 1
    row = W * nc * sb
    out = torch.empty((F, H, row), dtype=torch.uint8, device="cuda")
    own = os.environ.get("OWN_TABLES") == "1"  # per-frame tables in device memory (here: F copies of the same ones)
    flags = int(os.environ.get("FLAGS", "0"))
    wsb = api.workspace_bytes(info, F, flags, own_tables=own)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream()
    qd = 0
    if own:
        tabs = np.ones((F, 4, 64), np.uint16)
        for c in range(nc):
            tabs[:, c] = np.array(info.quant[info.quant_index[c]][:], np.uint16)
        qdev = torch.from_numpy(tabs.view(np.int16)).cuda()
        qd = qdev.data_ptr()

    def step():
        api.launch_reconstruct(info, coef.data_ptr(), out.data_ptr(), F, row, H * row, n, flags=flags, workspace=ws.data_ptr(), workspace_bytes=wsb,
                               stream=stream.cuda_stream, quant_dev=qd)

    for _ in range(20):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(20):
        step()
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    bpp = 2.0 * n / (W * H) + nc * sb  # int16 coefficients in, bytes out
    ok = "n/a"
    if d is not None:
        got = out[0].cpu().numpy()
        ok = bool(np.array_equal((got.view(np.uint16) if p12 else got).reshape(H, W, nc).squeeze(), d.reconstruct().squeeze()))
    print(f"{label:>8}: range_max {list(info.range_max)[:nc]} {api.kernel_name(info, flags):<44} {ms:7.3f} ms/launch {W*H*F/ms/1e6:8.1f} Gpixel/s "
          f"{W*H*F*bpp/ms/1e6:7.0f} GB/s algorithmic ({bpp:.1f} B/px, {W*H*F*bpp/ms/1e6/8000:.2f} of 8 TB/s) same as decoder object: {ok}{' (per-frame tables)' if own else ''}", flush=True)


for sub in os.environ.get("LAYOUTS", "420,444,422,440,gray").split(","):
    p12 = sub.endswith("_12")  # 12-bit frame: the 8-bit stream's entropy coded data with deltas times 16 (synth.to_12bit)
    sub = sub[:-3] if p12 else sub
    img = synth.synth_image(W, H, 1234, channels=1 if sub == "gray" else 3)
    if os.environ.get("SATURATED") == "1" and sub != "gray":
        # saturated graphics over the synthetic picture: hard-edged bars of pure colours push the chroma range check of the
        # frame past the packed gate (sum |c| q >= 2047 in some block), which selects the wide / 32-bit flavours
        img = img.copy()
        img[:, 1000:1400] = (255, 0, 0)
        img[:, 1400:1800] = (0, 0, 255)
        img[2000:2300] = (255, 255, 0)
        img[2300:2600] = (0, 255, 0)
        img[3000:3200:5, 3000:3400] = (255, 0, 255)   # thin lines, thin columns: strong chroma AC in every layout
        img[3000:3200, 3400:3800:5] = (0, 255, 255)
        img[3300:3500:6, 3000:3800:6] = (255, 0, 0)
    d = api.Decoder(0)
    if sub in HEADER_EDITS:
        # layouts no encoder here writes: a 4:4:4 stream (all planes full size) whose frame header gets other sampling factors,
        # decoded from its coefficients as such a frame -- plane sizes are recomputed by mijpeg_frame_layout, the planes are
        # filled with the 4:4:4 stream's luma blocks (content statistics of a photograph, which is all a rate needs)
        base = d.read(synth.encode_jpeg(img, 85, "444", restart_mcus=8))
        hs, vs = HEADER_EDITS[sub]
        info = api.frame_layout(W, H, 3, hs, vs, [list(base.quant[t]) for t in range(4)], quant_index=list(base.quant_index)[:3], ycbcr=0 if sub == "lumasub" else 1)
        info.fast_arith = 1
        info.sample_bytes = 1
        src_plane = torch.from_numpy(d.coefficients(0).reshape(-1).copy()).cuda()
        n = int(info.coef_count)
        one = torch.zeros(n, dtype=torch.int16, device="cuda")
        for c in range(3):
            nb = info.blocks_w[c] * info.blocks_h[c] * 64
            off = int(info.coef_offset[c])
            one[off:off + nb] = src_plane[:nb] if nb <= src_plane.numel() else src_plane.repeat((nb + src_plane.numel() - 1) // src_plane.numel())[:nb]
            info.range_max[c] = base.range_max[0]
        coef = one.unsqueeze(0).repeat(F, 1).contiguous()
        nc, p12 = 3, False
        run_layout(sub, info, coef, None)
        d.close()
        continue
    if sub == "cmyk":
        import io

        from PIL import Image
        buf = io.BytesIO()
        Image.fromarray(np.dstack([img, img[:, :, 1]]), "CMYK").save(buf, format="JPEG", quality=85)
        data = buf.getvalue()
    else:
        # Pillow has no 4:4:0: that stream comes from this library's own encoder
        data = d.encode(img, 85, sub, 8) if sub in ("440", "411") else synth.encode_jpeg(img, 85, "444" if sub == "gray" else sub, restart_mcus=8)
    if p12:
        data = synth.to_12bit(data)
    info = d.read(data)
    n = int(info.coef_count)
    coef = torch.empty((F, n), dtype=torch.int16, device="cuda")
    src = d.device_coefficients()
    torch.cuda.synchronize()
    for f in range(F):
        hip.hipMemcpy(C.c_void_p(coef[f].data_ptr()), C.c_void_p(src), C.c_size_t(n * 2), 3)
    run_layout(sub + ("_12" if p12 else ""), info, coef, d)
    d.close()
