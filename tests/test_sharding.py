"""Multi-rank path on CPU: two processes over the gloo backend run the same image-sharding logic bench.py and
the batch config use (independent frames per rank, barrier on both sides of the timed region, MAX over ranks of
the elapsed time, SUM of the work) -- with the host entropy decoder as the per-rank work, since there is no GPU
here.  There is no data-path collective to test: frames never move between ranks."""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

WORKER = textwrap.dedent("""
    import os, sys, time, hashlib
    sys.path.insert(0, %r)
    import numpy as np
    import torch
    import torch.distributed as dist
    from libjpeg_amd import api, synth, sharding

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    frames = list(range(10))                       # global frame ids of the batch
    mine = sharding.frames_of_rank(len(frames), rank, world)
    assert sorted(sum((sharding.frames_of_rank(len(frames), r, world) for r in range(world)), [])) == frames
    dec = api.Decoder(None)                        # host-only: parsing + Huffman decoding
    digest = hashlib.sha256()
    def work():
        for i in mine:
            dec.read(synth.synth_jpeg(160, 96, seed=1000 + i, quality=85, subsampling="420", restart_mcus=2), threads=1)
            digest.update(dec.coefficients(0).tobytes())
        return len(mine) * 160 * 96
    elapsed, units = sharding.timed_region(work, dist)
    total_units, max_elapsed = sharding.reduce_result(units, elapsed, dist)
    assert total_units == 10 * 160 * 96, total_units
    assert max_elapsed >= elapsed - 1e-9
    print("RANK", rank, "frames", mine, "units", units, "total", total_units, flush=True)
    dist.barrier()
    dist.destroy_process_group()
""")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_sharding_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert "frames [0, 2, 4, 6, 8]" in outs[0] and "frames [1, 3, 5, 7, 9]" in outs[1]


def test_frames_of_rank_partitions():
    from libjpeg_amd import sharding

    for n in (0, 1, 7, 256):
        for world in (1, 2, 4, 8):
            parts = [sharding.frames_of_rank(n, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


BATCH_WORKER = textwrap.dedent("""
    import os, sys, hashlib
    sys.path.insert(0, %r)
    import numpy as np
    import torch.distributed as dist
    from libjpeg_amd import batch
    from oracle import oracle as O

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    cfg = dict(batch.CONFIG4, width=208, height=112, frames=12)    # config 4's recipe at a size the CPU suite can afford
    streams = batch.make_streams(range(cfg["frames"]), cfg, workers=1)
    r = batch.run_sharded(streams, cfg["frames"], rank, world, None, dist, steps=2, warmup=1)   # device=None: host entropy decoder
    assert r["total_pixels"] == 2 * cfg["frames"] * cfg["width"] * cfg["height"], r["total_pixels"]
    assert r["frames"] == list(range(rank, cfg["frames"], world))
    shard = r["shard"]
    for k, i in enumerate(r["frames"]):
        info, planes = O.decode_coefficients(streams[i])
        for c in range(3):
            assert np.array_equal(shard.coefficients[k][c], planes[c]), (i, c)
    shard.close()
    print("RANK", rank, "frames", r["frames"], "ok", flush=True)
    dist.barrier()
    dist.destroy_process_group()
""")


def test_config4_driver_two_ranks_gloo(tmp_path):
    """The batch driver of BASELINE config 4 (libjpeg_amd/batch.run_sharded: frames r, r+N, ..., barriers around the timed
    region, SUM of the pixels, MAX of the time) on two gloo ranks with the host entropy decoder as the per-rank work; every
    frame's coefficients equal the oracle's."""
    script = tmp_path / "batch_worker.py"
    script.write_text(BATCH_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2", MIJPEG_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert "frames [0, 2, 4, 6, 8, 10] ok" in outs[0] and "frames [1, 3, 5, 7, 9, 11] ok" in outs[1]


def test_host_threads_are_divided_among_the_ranks():
    from libjpeg_amd import batch

    cores = os.cpu_count() or 1
    assert batch.host_threads(1) == min(64, cores)
    assert batch.host_threads(8) == max(1, min(64, cores // 8))
    assert batch.host_threads(8) * 8 <= max(cores, 8)


def test_bind_to_gpu_node_is_a_hint_that_never_fails():
    """Without a GPU (or without sysfs topology) nothing happens and nothing is raised; the affinity stays what it was."""
    import os

    from libjpeg_amd import sharding
    before = os.sched_getaffinity(0)
    assert sharding.bind_to_gpu_node(0) is None
    assert os.sched_getaffinity(0) == before


def test_batch_schedule_covers_every_frame_once():
    """BatchShard.schedule: contiguous chunks that cover the shard, at most `chunk` frames each; the ramped schedule starts and
    ends with smaller chunks (the pipeline's fill and drain) and is the plain one for shards too small to ramp."""
    from libjpeg_amd.batch import BatchShard
    for n in (1, 7, 32, 100, 256, 1000):
        for chunk in (1, 8, 24, 32, 40):
            for ramp in (False, True):
                s = BatchShard.schedule(n, chunk, ramp)
                assert s[0][0] == 0 and s[-1][1] == n
                assert all(a[1] == b[0] for a, b in zip(s, s[1:]))
                assert all(0 < b - a <= chunk for a, b in s)
    r = BatchShard.schedule(256, 24, True)
    sizes = [b - a for a, b in r]
    assert sizes[0] < sizes[1] < sizes[2] and sizes[-1] < sizes[-2] < sizes[-3] and sizes[:2] == sizes[-2:][::-1]
    assert BatchShard.schedule(32, 24, True) == BatchShard.schedule(32, 24, False)


def _bench(*args, env=None, timeout=900):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout,
                          env=dict(os.environ, **(env or {})))


def test_bench_never_reports_more_gpus_than_it_runs_on():
    """`bench.py --gpus N` starts its ranks itself; what it cannot start is an error, not a run on fewer devices."""
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 64:
        pytest.skip("a box with 64 devices")
    r = _bench("--gpus", "64", "--steps", "1")
    assert r.returncode != 0 and "--gpus 64" in r.stderr and "visible" in r.stderr, r.stderr[-400:]
    r = _bench("--gpus", "4", "--steps", "1", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr, r.stderr[-400:]
    r = _bench("--gpus", "2", "--launcher", "never", "--steps", "1")
    assert r.returncode != 0 and "refusing" in r.stderr, r.stderr[-400:]


@pytest.mark.gpu
def test_bench_through_its_own_launcher_over_rccl():
    """One rank, started the way N ranks are (`--launcher always`): init_process_group("nccl"), the barriers around both timed regions
    and the reductions run on the device over RCCL; the line says so, and every rank's frames are checked against the oracle."""
    import json

    r = _bench("--gpus", "1", "--launcher", "always", "--workload", "batch4k", "--batch-frames", "32", "--batch-steps", "2", "--frames", "8",
               "--steps", "2", "--warmup", "1", "--settle-ms", "0", "--no-traffic", "--no-dense", "--no-xt", "--no-end-to-end", "--no-cpu-baseline",
               "--emulate-world", "0", env={"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-600:], r.stderr[-1200:])
    res = json.loads(lines[0])
    assert res["n_gpus"] == 1 and res["verified"] is True
    assert res["launch"] == {"launched_by": "torch.distributed.run", "world": 1, "collectives": "nccl", "visible_devices": res["launch"]["visible_devices"]}
    b = res["batch4k"]
    assert b["verified"] is True and b["per_rank_verified"] == [True] and b["n_gpus"] == 1 and b["frames"] == 32
    assert len(b["per_rank_ms"]) == 1 and b["per_rank_host_threads"] == [b["host_threads_per_rank"]]
