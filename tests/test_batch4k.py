"""BASELINE config 4 at its stated shape: 256 independent 3840x2160 4:2:0 Q85 DRI=8 frames (seeds 1000...1255), compressed
streams in host memory -> pixels in HBM through the batch entry points of the C ABI, every frame against the oracle."""
import hashlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from libjpeg_amd import batch


@pytest.mark.gpu
def test_config4_all_256_frames_against_the_oracle(oracle):
    cfg = batch.CONFIG4
    n = cfg["frames"]
    streams = batch.make_streams(range(n), cfg)
    assert len({hashlib.sha256(s).digest() for s in streams.values()}) == n  # 256 distinct pictures
    r = batch.run_sharded(streams, n, 0, 1, 0, None, steps=1, warmup=0, chunk=32, depth=2)
    shard = r["shard"]
    assert r["total_pixels"] == n * cfg["width"] * cfg["height"]

    def check(i):
        exp = oracle.decode(streams[i])  # ctypes releases the GIL: the oracle decodes run on the host cores in parallel
        got = shard.pixels(i)
        return i if not np.array_equal(got, exp) else -1

    with ThreadPoolExecutor(32) as ex:
        bad = [i for i in ex.map(check, range(n)) if i >= 0]
    shard.close()
    assert not bad, f"frames differing from the oracle: {bad[:10]}"


@pytest.mark.gpu
def test_config4_shards_are_what_the_whole_batch_is(oracle):
    """Rank r of 4 decodes frames r, r+4, ...: the shard's frames equal the oracle's decode of exactly those streams (a
    smaller batch of the same shape keeps this fast)."""
    cfg = dict(batch.CONFIG4, frames=16)
    streams = batch.make_streams(range(16), cfg)
    for rank in (1, 3):
        r = batch.run_sharded(streams, 16, rank, 4, 0, None, steps=1, chunk=3, depth=2)
        assert r["frames"] == list(range(rank, 16, 4))
        for k, i in enumerate(r["frames"]):
            assert np.array_equal(r["shard"].pixels(k), oracle.decode(streams[i])), (rank, i)
        r["shard"].close()
