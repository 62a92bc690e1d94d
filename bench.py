#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X JPEG block-decode path (BASELINE.json metric).

A "step" is one pass of the hot path (fused dequant + IDCT + chroma upsample + YCbCr->RGB kernel) over one
batch of F synthetic 8K (7680x4320) 4:2:0 baseline frames whose int16 coefficient planes are already
resident in HBM; pixels are written to HBM.  value = decoded Mpixels/s over all ranks.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F] [--size 8k|4k]

N > 1 is launched by the driver through torch.distributed.run (one rank per GPU, RCCL barrier only: frames
are independent, nothing is exchanged).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from libjpeg_amd import api, sharding, synth  # noqa: E402

SIZES = {"8k": (7680, 4320), "4k": (3840, 2160)}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_PIXEL_420 = 6.0  # int16 coefficients in (3 B/px) + interleaved RGB out (3 B/px)


def cpu_baseline(jpeg_bytes, width, height):
    """Reference CPU path timed on this box's host cores (rank 0, N=1 only).  Uses oracle/_ref/jpeg (the
    real reference binary, kind 'reference') if it was built, else the oracle's C restatement ('port')."""
    from oracle import oracle as O

    mpix = width * height / 1e6
    if O.have_reference():
        tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else None
        with tempfile.TemporaryDirectory(dir=tmpdir) as d:
            src, dst = os.path.join(d, "in.jpg"), os.path.join(d, "out.ppm")
            with open(src, "wb") as f:
                f.write(jpeg_bytes)
            ts = []
            for _ in range(5):
                t = time.perf_counter()
                subprocess.run([O.REF_BIN, src, dst], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                ts.append(time.perf_counter() - t)
        best = sorted(ts)[len(ts) // 2]
        return dict(value=round(mpix / best, 2), unit="Mpixels/s", cores=1, kind="reference",
                    sample=f"5 whole-process decodes of one {width}x{height} 4:2:0 Q85 DRI=8 frame by oracle/_ref/jpeg "
                           f"(file in /dev/shm -> PPM in /dev/shm), median {best * 1e3:.0f} ms")
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        O.decode(jpeg_bytes)
        ts.append(time.perf_counter() - t)
    best = sorted(ts)[len(ts) // 2]
    return dict(value=round(mpix / best, 2), unit="Mpixels/s", cores=1, kind="port",
                sample=f"5 in-memory decodes of one {width}x{height} 4:2:0 frame by oracle/liboracle.so, median {best * 1e3:.0f} ms")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=64, help="frames per step (= one kernel launch) and rank, device resident")
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="untimed launches before the warmup steps: MI355X power management needs ~50 ms of sustained load "
                         "before the clock settles (profiles/r01/step_times.txt); 0 disables")
    ap.add_argument("--size", default="8k", choices=sorted(SIZES))
    ap.add_argument("--subsampling", default="420", choices=["420", "444"], help="420 = the BASELINE workload; 444 for side measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the reconstruction path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_

        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    W, H = SIZES[args.size]
    F = args.frames
    # ---- workload: F frames per rank, two distinct pictures per rank, coefficient planes in HBM ----------
    dec = api.Decoder(local_rank)
    host_planes, jpegs = [], []
    info = None
    for i in range(2):
        data = synth.synth_jpeg(W, H, seed=1234 + 17 * rank + i, quality=85, subsampling=args.subsampling, restart_mcus=8)
        jpegs.append(data)
        info = dec.read(data)
        host_planes.append(np.concatenate([dec.coefficients(c).reshape(-1) for c in range(info.components)]))
    n = int(info.coef_count)
    coef = torch.empty((F, n), dtype=torch.int16, device="cuda")
    for f in range(F):
        if f < 2:
            coef[f].copy_(torch.from_numpy(host_planes[f]))
        else:
            coef[f].copy_(coef[f % 2])
    row = W * 3
    out = torch.empty((F, H, row), dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream()

    def step():
        api.launch_reconstruct(info, coef.data_ptr(), out.data_ptr(), F, row, H * row, n, stream=stream.cuda_stream)

    # bring the device to its steady-state clock (untimed, disclosed in config.settle_launches)
    settle_launches = 0
    if args.settle_ms > 0:
        torch.cuda.synchronize()
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
            for _ in range(4):
                step()
            settle_launches += 4
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed_steps():
        ev0.record(stream)
        for _ in range(args.steps):
            step()
        ev1.record(stream)
        return W * H * F * args.steps

    # barrier + torch.cuda.synchronize() on both sides, MAX over ranks of the elapsed time (libjpeg_amd/sharding.py)
    wall, pixels = sharding.timed_region(timed_steps, dist)
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream: one kernel per step
    total_pixels, wall = sharding.reduce_result(pixels, wall, dist)
    if dist:
        t = torch.tensor([kernel_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        kernel_ms = float(t[0])
    assert total_pixels == W * H * F * args.steps * world

    pixels_per_step = W * H * F * world
    ms_per_step = wall * 1e3 / args.steps
    value = pixels_per_step / (ms_per_step * 1e-3) / 1e6
    bpp = BYTES_PER_PIXEL_420 if args.subsampling == "420" else 9.0  # 4:4:4: 3 x 2 B in + 3 B out
    alg_bytes = W * H * F * bpp  # per launch (one rank)
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9

    result = {
        "metric": "decoded Mpixels/s, 8K 4:2:0 baseline (fused IDCT+upsample+YCbCr kernel, coefficients resident in HBM)",
        "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32 (int16 coefficients in, u8 pixels out)", "data": "synthetic",
        "config": {"workload": f"{F} x {W}x{H} {args.subsampling[0]}:{args.subsampling[1]}:{args.subsampling[2]} Q85 DRI=8 baseline frames per GPU per step (BASELINE configs[2] frame shape, "
                               f"device-resident coefficient planes)", "frames_per_gpu": F, "kernel": api.kernel_name(info), "settle_launches": settle_launches,
                   "fast_arith": int(info.fast_arith), "parallelism": f"image-sharded x{world}, no data-path collective"},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                     "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": int(alg_bytes)},
    }

    # HBM bytes per launch measured by the PMC passes (tools/gpu_profile.sh): counters cannot be read from inside this
    # process, so the number measured for this exact workload is carried in profiles/ next to the CSVs it came from
    try:
        with open(os.path.join(ROOT, "profiles", "r01", "traffic.json")) as f:
            t = json.load(f).get(api.kernel_name(info))
        if t and (t["width"], t["height"]) == (W, H):
            result["roofline"]["traffic"] = int(t["traffic_bytes"] * F / t["frames"])  # counters were collected on t["frames"] frames per launch
    except (OSError, ValueError, KeyError):
        pass

    if rank == 0 and not args.no_end_to_end:
        # whole decode of one frame through the decoder object: host Huffman (all cores) + streaming H2D +
        # kernel + D2H into host memory.  PCIe/host inclusive -- reported beside, never as `value`.
        user = np.empty((H, W, 3), np.uint8)
        user[:] = 0  # touch the pages once: a real client reuses its frame buffer
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            dec.read(jpegs[0])
            dec.reconstruct(out=user)
            ts.append(time.perf_counter() - t)
        tm = dec.timing()
        best = min(ts)
        result["end_to_end"] = {"value": round(W * H / best / 1e6, 1), "unit": "Mpixels/s", "ms": round(best * 1e3, 2),
                                "host_threads": api.default_threads(), "host_cores": os.cpu_count(),
                                "phases_ms": {k: round(v * 1e3, 2) for k, v in tm.items()},
                                "note": "one frame, PCIe and host inclusive: bytes -> host Huffman (restart-interval parallel) -> "
                                        "pinned H2D (streamed) -> kernel -> D2H -> copy into the caller's interleaved bitmap"}
    if rank == 0 and not args.no_end_to_end:
        # the same, pipelined over a batch of frames (config 4's end-to-end shape): two decoder objects / streams
        from libjpeg_amd import pipeline

        nb = 24
        pipe = pipeline.FramePipeline(local_rank, depth=2, entropy="host")
        pipe.run([jpegs[i % 2] for i in range(4)])  # warm: pinned buffers, worker threads
        t = time.perf_counter()
        pipe.run([jpegs[i % 2] for i in range(nb)])
        dt = time.perf_counter() - t
        pipe.close()
        result["end_to_end"]["pipelined"] = {"value": round(W * H * nb / dt / 1e6, 1), "unit": "Mpixels/s", "frames": nb,
                                             "ms_per_frame": round(dt / nb * 1e3, 2), "depth": 2,
                                             "note": "two decoder objects / streams, D2H straight into pinned frames; bound by the ~200 MB per 8K frame that cross PCIe"}
    if rank == 0 and not args.no_end_to_end:
        # the same with the entropy decoding on the device (restart intervals in parallel, csrc/huffman.hip): only the
        # compressed bytes go up.  One frame to host memory, one frame left in HBM, and the two-deep pipeline.
        ts, tr = [], []
        for _ in range(5):
            t = time.perf_counter()
            dec.read(jpegs[0], entropy="gpu")
            t1 = time.perf_counter()
            dec.reconstruct(out=user)
            ts.append(time.perf_counter() - t)
            tr.append(t1 - t)
        dev_out = torch.empty((H, row), dtype=torch.uint8, device="cuda")
        th = []
        for _ in range(5):
            t = time.perf_counter()
            dec.read(jpegs[0], entropy="gpu")
            dec.reconstruct_device(dev_out.data_ptr(), row)
            th.append(time.perf_counter() - t)
        best_dt, best_depth = None, 0
        for depth in (2, 3):
            pipe = pipeline.FramePipeline(local_rank, depth=depth, entropy="gpu")
            pipe.run([jpegs[i % 2] for i in range(4)])
            t = time.perf_counter()
            pipe.run([jpegs[i % 2] for i in range(nb)])
            dt = time.perf_counter() - t
            pipe.close()
            if best_dt is None or dt < best_dt:
                best_dt, best_depth = dt, depth
        dt = best_dt
        # batch of frames: one Huffman launch + one reconstruction launch, pixels left in HBM (mijpeg_decode_batch_device)
        nbatch = 32
        bdec = api.Decoder(local_rank)
        bout = torch.empty((nbatch, H, row), dtype=torch.uint8, device="cuda")
        tb = []
        for _ in range(4):
            t = time.perf_counter()
            bdec.decode_batch_device([jpegs[i % 2] for i in range(nbatch)])
            bdec.reconstruct_batch_device(bout.data_ptr(), H * row, row)
            tb.append(time.perf_counter() - t)
        # the same streams without restart markers: the device finds virtual restart points itself (huffman_walk_kernel)
        plain = [synth.synth_jpeg(W, H, seed=1234 + 17 * rank + i, quality=85, subsampling=args.subsampling, restart_mcus=0) for i in range(2)]
        tn = []
        for _ in range(5):
            t = time.perf_counter()
            dec.read(plain[0], entropy="gpu")
            dec.reconstruct_device(dev_out.data_ptr(), row)
            tn.append(time.perf_counter() - t)
        walk_rounds = dec.device_walk_rounds()
        nplain = 16
        tnb = []
        for _ in range(3):
            t = time.perf_counter()
            bdec.decode_batch_device([plain[i % 2] for i in range(nplain)])
            bdec.reconstruct_batch_device(bout.data_ptr(), H * row, row)
            tnb.append(time.perf_counter() - t)
        bdec.close()
        del bout
        result["end_to_end"]["device_entropy_no_restart_markers"] = {
            "pixels_left_in_hbm_ms": round(min(tn) * 1e3, 2), "value": round(W * H / min(tn) / 1e6, 1), "unit": "Mpixels/s", "walk_rounds": walk_rounds,
            "batch": {"frames": nplain, "ms_per_frame": round(min(tnb) / nplain * 1e3, 3), "value": round(W * H * nplain / min(tnb) / 1e6, 1), "unit": "Mpixels/s"},
            "stream_bytes": len(plain[0]),
            "note": "no DRI: huffman_walk_kernel rounds to a fixed point of the subsequence hand-over states, prefix sums, virtual restart "
                    "intervals emitted on the device, then huffman_scan_kernel and the fused kernel; no host thread decodes"}
        result["end_to_end"]["device_entropy"] = {
            "batch": {"frames": nbatch, "ms_per_frame": round(min(tb) / nbatch * 1e3, 3), "value": round(W * H * nbatch / min(tb) / 1e6, 1),
                      "unit": "Mpixels/s", "note": "32 streams in host memory -> parallel header parse -> H2D of the compressed bytes -> one "
                                                   "huffman_scan_kernel launch -> one fused kernel launch, pixels left in HBM"},
            "value": round(W * H / min(ts) / 1e6, 1), "unit": "Mpixels/s", "ms": round(min(ts) * 1e3, 2),
            "read_ms": round(min(tr) * 1e3, 2), "pixels_left_in_hbm_ms": round(min(th) * 1e3, 2),
            "pipelined_ms_per_frame": round(dt / nb * 1e3, 2), "pipelined_value": round(W * H * nb / dt / 1e6, 1),
            "pipelined_depth": best_depth,
            "restart_interval_mcus": 8, "stream_bytes": len(jpegs[0]),
            "note": "bytes -> header parse + restart marker search on the host -> H2D of the compressed stream -> "
                    "huffman_scan_kernel (one lane per restart interval) -> fused kernel -> D2H of the pixels"}
    if rank == 0 and not args.no_end_to_end:
        # encoder direction of the block pipeline: forward kernels on frames resident in HBM, and one picture
        # from host memory to a baseline stream (upload, kernels, download of the coefficients, entropy coder on the host)
        try:
            img = synth.synth_image(W, H, 1234 + 17 * rank)
            fi = api.frame_layout(W, H, 3, (2, 1, 1), (2, 1, 1), [list(info.quant[t]) for t in range(4)], quant_index=list(info.quant_index)[:3])
            FE = 8
            px = torch.from_numpy(img).cuda().unsqueeze(0).repeat(FE, 1, 1, 1).contiguous()
            fcoef = torch.empty((FE, int(fi.coef_count)), dtype=torch.int16, device="cuda")
            for _ in range(150):  # settles the clocks like the main measurement does
                api.launch_forward(fi, px.data_ptr(), fcoef.data_ptr(), FE, W * 3, H * W * 3, stream=stream.cuda_stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(20):
                api.launch_forward(fi, px.data_ptr(), fcoef.data_ptr(), FE, W * 3, H * W * 3, stream=stream.cuda_stream)
            e1.record(stream)
            torch.cuda.synchronize()
            fms = e0.elapsed_time(e1) / 20
            # the frames stay where they are: forward kernels for all of them, device entropy coder frame by frame
            dec.encode_batch_device(fi, px.data_ptr(), fcoef.data_ptr(), 2, W * 3, H * W * 3, 8, False)
            tb = []
            for _ in range(3):
                streams = dec.encode_batch_device(fi, px.data_ptr(), fcoef.data_ptr(), FE, W * 3, H * W * 3, 8, False)
                tb.append(list(dec.timing().values())[0])  # the C call (the Python wrapper copies the streams once more)
            del px, fcoef
            dec.encode(img, 85, "420", 8, False)
            te = []
            for _ in range(4):
                t = time.perf_counter()
                stream_bytes = dec.encode(img, 85, "420", 8, False)
                te.append(time.perf_counter() - t)
            result["end_to_end"]["encoder_direction"] = {
                "forward_kernels": {"value": round(W * H * FE / fms / 1e3, 1), "unit": "Mpixels/s", "ms": round(fms, 3), "frames": FE,
                                    "algorithmic_GBps": round(W * H * FE * 6 / fms / 1e6, 1),
                                    "note": "RGB in HBM -> YCbCr 4:2:0 -> FDCT -> quantiser -> int16 planes in HBM (fdct420_tile_kernel + fdct_blocks_kernel for the edges)"},
                "encode_frames_in_hbm": {"value": round(W * H * FE / min(tb) / 1e6, 1), "unit": "Mpixels/s", "ms_per_frame": round(min(tb) / FE * 1e3, 3), "frames": FE,
                                         "stream_bytes": len(streams[0]),
                                         "note": "frames resident in HBM -> baseline JPEG streams in host memory: forward kernels + on-device entropy "
                                                 "coder (mijpeg_encode_batch_device); only the streams cross PCIe"},
                "encode_picture": {"value": round(W * H / min(te) / 1e6, 1), "unit": "Mpixels/s", "ms": round(min(te) * 1e3, 2), "stream_bytes": len(stream_bytes),
                                   "note": "one 8K picture in host memory -> baseline JPEG, restart interval 8, Annex K tables: the reference "
                                           "encoder's tables and coefficients (mijpeg_encode_image)"}}
        except Exception as e:  # a side measurement never costs the headline number
            result["end_to_end"]["encoder_direction"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(jpegs[0], W, H)
        except Exception as e:  # the baseline is a report, never a reason to lose the GPU number
            result["cpu_baseline"] = {"value": None, "error": repr(e)}
    dec.close()
    if rank == 0:
        print(json.dumps(result))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
