#!/usr/bin/env python3
"""Config 5's reconstruction launches alone, for the profiler: the 4K HDR stream of bench.py's xt_profile_c (`-r12`, or with
`--hidden` four hidden residual bits: fusedxtw420_kernel), decoded once, then N launches on 8 device-resident frames.
    rocprofv3 --kernel-trace --stats -- python tools/xt_launches.py --hidden"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from libjpeg_amd import api, synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", action="store_true")
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--time", action="store_true")
    a = ap.parse_args()
    W, H = bench.SIZES["4k"]
    data = O.reference_encode_hdr(synth.synth_hdr(W, H, 99), bench.XT_ARGS + (["-rR", "4"] if a.hidden else []))
    dec = api.Decoder(0)
    info = dec.read(data, entropy="host")
    xt = dec.xt_params()
    n, F = int(info.coef_count), a.frames
    coef = torch.empty((F, n), dtype=torch.int16, device="cuda")
    src = dec.device_coefficients()
    dec.synchronize()
    torch.cuda.synchronize()
    hip = C.CDLL("libamdhip64.so")
    for f in range(F):
        assert hip.hipMemcpy(C.c_void_p(coef[f].data_ptr()), C.c_void_p(src), C.c_size_t(n * 2), 3) == 0
    row = W * 3 * 2
    out = torch.empty((F, H, row), dtype=torch.uint8, device="cuda")
    wsb = api.workspace_bytes(info, F, xt=xt)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()

    def launch():
        api.launch_reconstruct(info, coef.data_ptr(), out.data_ptr(), F, row, H * row, n, workspace=ws.data_ptr(), workspace_bytes=wsb, stream=s.cuda_stream, xt=xt)

    if a.time:  # settle the clocks, then HIP events around the launches (on the launch stream)
        import time
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2:
            launch()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
    for _ in range(a.launches):
        launch()
    if a.time:
        e1.record(s)
    torch.cuda.synchronize()
    print(api.kernel_name(info, xt=xt), "launches", a.launches, "frames", F, ("ms per launch %.4f" % (e0.elapsed_time(e1) / a.launches)) if a.time else "")


if __name__ == "__main__":
    main()
