#!/usr/bin/env python3
"""BASELINE config 4 on one GPU without the rest of bench.py (for rocprofv3 runs): N frames of 4K 4:2:0 Q85 DRI=8
(CFG_FRAMES, default 128), bytes in host memory -> pixels in HBM; prints ms per batch for a few chunk / depth settings."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from libjpeg_amd import batch, sharding  # noqa: E402


def main():
    if not os.environ.get("MIJPEG_BENCH_NO_NUMA"):
        print("NUMA binding:", sharding.bind_to_gpu_node(0))  # like bench.py: this process on the socket its GPU hangs off
    n = int(os.environ.get("CFG_FRAMES", "128"))
    dri = int(os.environ.get("CFG_DRI", "8"))
    cfg = dict(batch.CONFIG4, frames=n, restart_mcus=dri, quality=int(os.environ.get("CFG_Q", batch.CONFIG4["quality"])))
    if os.environ.get("CFG_8K"):
        cfg.update(width=7680, height=4320)
    t = time.perf_counter()
    streams = batch.make_streams(range(n), cfg)
    print(f"generated {n} streams in {time.perf_counter() - t:.1f} s, {sum(len(s) for s in streams.values()) / n / 1e6:.2f} MB each")
    # CHUNKxDEPTH, a trailing r = ramped schedule (small chunks at both ends)
    settings = [(int(c), int(d.rstrip("r")), d.endswith("r")) for c, d in (s.split("x") for s in os.environ.get("SETTINGS", "32x2,16x3,64x2,32x1,128x1").split(","))]
    if os.environ.get("DOWNLOAD"):
        # DOWNLOAD=1: pixels into pinned host memory, each chunk's download under the decode of the next ones (BatchShard.run(download_to))
        import time as _t
        for chunk, depth, ramp in settings:
            shard = batch.BatchShard([streams[i] for i in range(n)], 0, chunk, depth, ramp)
            host = torch.empty(shard.out.shape, dtype=shard.out.dtype).pin_memory()
            shard.run(download_to=host)
            ts = []
            for _ in range(int(os.environ.get("STEPS", "3"))):
                torch.cuda.synchronize(); t = _t.perf_counter(); shard.run(download_to=host); ts.append((_t.perf_counter() - t) * 1e3)
            print(f"chunk {chunk:4d} depth {depth}{' ramp' if ramp else '     '}: bytes -> pixels in pinned host memory {min(ts):8.2f} ms per batch ({host.numel() / 1e9 / (min(ts) * 1e-3):.1f} GB/s down)")
            shard.close()
        return
    for chunk, depth, ramp in settings:
        r = batch.run_sharded(streams, n, 0, 1, 0, None, steps=int(os.environ.get("STEPS", "3")), warmup=1, chunk=chunk, depth=depth, ramp=ramp)
        ms = r["seconds"] * 1e3 / int(os.environ.get("STEPS", "3"))
        px = cfg["width"] * cfg["height"] * n
        print(f"chunk {chunk:4d} depth {depth}{' ramp' if ramp else '     '}: {ms:8.2f} ms per batch, {ms / n:.4f} ms per frame, {px / ms / 1e6:8.1f} Gpixel/s   chunk ms {['%.1f' % x for x in r['shard'].chunk_ms[:6]]}")
        ph = [p for p in getattr(r["shard"], "chunk_phases_ms", []) if p]
        if ph:
            print("      decode call / parse / prepare / upload+huffman+status ms, mean over chunks:", ["%.2f" % (sum(p[k] for p in ph) / len(ph)) for k in range(4)])
        r["shard"].close()


if __name__ == "__main__":  # the stream generator spawns worker processes that import this file
    main()
