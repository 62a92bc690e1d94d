#!/usr/bin/env python3
"""Config 4's stall mode: neighbouring (chunk, objects) settings of the batch pipeline that take 1.4-2x the
best.  Every step of every setting is timed on its own (synchronised), with the host time of every chunk's submit call and its
phases, so that a stall shows as WHAT it is: one step, one chunk, one phase.
    CFG_FRAMES=256 SETTINGS=24x4,24x4r,32x4,11x3 STEPS=12 python tools/batch_stall_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from libjpeg_amd import batch, sharding  # noqa: E402


def gpu_clock():
    """current shader clock in MHz (sysfs pp_dpm_sclk of the first AMD card: the line with the asterisk), None where it cannot be read"""
    import glob

    for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
        try:
            for line in open(f):
                if "*" in line:
                    return int(line.split(":")[1].strip().split("M")[0])
        except Exception:  # noqa: BLE001
            pass
    return None


def throttled():
    """(periods in which the container's CPU quota ran out, microseconds its threads stood still) so far -- cgroup v2 cpu.stat;
    (None, None) where there is none"""
    try:
        d = dict(line.split() for line in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
    except (OSError, ValueError):
        return None, None


def main():
    if not os.environ.get("MIJPEG_BENCH_NO_NUMA"):
        print("NUMA binding:", sharding.bind_to_gpu_node(0))
    n = int(os.environ.get("CFG_FRAMES", "256"))
    cfg = dict(batch.CONFIG4, frames=n)
    streams = batch.make_streams(range(n), cfg)
    steps = int(os.environ.get("STEPS", "12"))
    settings = [(int(c), int(d.rstrip("r")), d.endswith("r")) for c, d in (s.split("x") for s in os.environ.get("SETTINGS", "24x4,24x4r,32x4").split(","))]
    for chunk, depth, ramp in settings:
        shard = batch.BatchShard([streams[i] for i in range(n)], 0, chunk, depth, ramp)
        for _ in range(int(os.environ.get("WARMUP", "2"))):
            shard.run()
        rows = []
        idle = float(os.environ.get("IDLE_MS", "0")) * 1e-3  # the device left alone between the steps
        for _ in range(steps):
            torch.cuda.synchronize()
            if idle:
                time.sleep(idle)
            sclk = gpu_clock()
            th0 = throttled()
            t = time.perf_counter()
            shard.run()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) * 1e3
            th1 = throttled()
            rows.append((dt, list(shard.chunk_ms), list(getattr(shard, "chunk_phases_ms", [])), sclk, gpu_clock(),
                         None if th0[0] is None else (th1[0] - th0[0], (th1[1] - th0[1]) / 1e3)))
        ms = sorted(r[0] for r in rows)
        print(f"chunk {chunk:3d} x {depth}{' ramp' if ramp else ''}: steps ms {' '.join('%.2f' % r[0] for r in rows)}   median {ms[len(ms) // 2]:.2f} min {ms[0]:.2f} max {ms[-1]:.2f}")
        best = ms[0]
        if rows[0][3] is not None:
            print('   shader clock MHz before/after each step:', ' '.join(f'{r[3]}/{r[4]}' for r in rows))
        if rows[0][5] is not None:  # the container's CPU quota (cgroup cpu.max) ran out inside a step?
            print("   quota periods that ran out inside each step / ms its threads stood still (summed over the threads):",
                  " ".join(f"{r[5][0]}/{r[5][1]:.0f}" for r in rows))
        for si, (tot, cms, ph, _, _, _) in enumerate(rows):
            if tot > 1.25 * best:
                worst = max(range(len(cms)), key=lambda i: cms[i])
                print(f"   step {si}: {tot:.2f} ms; submit ms per chunk {' '.join('%.2f' % x for x in cms)}")
                print(f"      slowest submit = chunk {worst}: (parse, h2d wait, kernel, d2h) phases ms {['%.2f' % x for x in ph[worst]] if ph[worst] else None}")
        shard.close()


if __name__ == "__main__":
    main()
