"""BASELINE config 4 at its stated shape: 256 independent 3840x2160 4:2:0 Q85 DRI=8 frames (seeds 1000...1255), compressed
streams in host memory -> pixels in HBM through the batch entry points of the C ABI, every frame against the oracle."""
import hashlib
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from conftest import ROOT
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


@pytest.mark.gpu
def test_submit_finish_pipeline_semantics(oracle):
    """mijpeg_submit_batch_device / mijpeg_finish_batch_device: the two halves give what the one call gives; a batch nobody
    waited for is waited for by the next submit; what the Huffman kernel reports arrives with finish; batches without restart
    markers (device walk: needs the host between its rounds) are complete when submit returns."""
    import torch

    from libjpeg_amd import api, synth

    w, h = 416, 240
    good = [synth.synth_jpeg(w, h, 300 + i, 85, "420", 2) for i in range(6)]
    exp = [oracle.decode(s) for s in good]
    row = w * 3
    out = torch.zeros((6, h, row), dtype=torch.uint8, device="cuda")
    d = api.Decoder(0)
    # 1. submit, then reconstruct (which finishes it)
    d.submit_batch_device(good, 1)
    d.reconstruct_batch_device(out.data_ptr(), h * row, row)
    for i in range(6):
        assert np.array_equal(out[i].cpu().numpy().reshape(h, w, 3), exp[i])
    # 2. submit twice in a row: the first batch is simply superseded
    d.submit_batch_device(good[:3], 1)
    d.submit_batch_device(good[3:], 1)
    info = d.finish_batch_device()
    assert info.width == w and d.batch_frames == 3
    out.zero_()
    d.reconstruct_batch_device(out.data_ptr(), h * row, row)
    for i in range(3):
        assert np.array_equal(out[i].cpu().numpy().reshape(h, w, 3), exp[3 + i])
    # 3. finish without anything submitted: an error, not a crash
    d2 = api.Decoder(0)
    with pytest.raises(api.MijpegError) as e:
        d2.finish_batch_device()
    assert e.value.code == -1031
    # 4. a member whose entropy coded data cannot be decoded inside an interval (a run of one-bits: no Huffman code is all
    #    ones): submit cannot know, finish reports NOT_AVAILABLE -- the caller decodes such a stream on its own and gets the
    #    reference's verdict from the host walk
    bad = bytearray(good[2])
    pos = bad.find(b"\xff\xda") + 300
    while any(x == 0xFF for x in bad[pos - 1:pos + 14]):
        pos += 1
    bad[pos:pos + 12] = b"\xff\x00" * 6
    d2.submit_batch_device([good[0], bytes(bad), good[1]], 1)
    with pytest.raises(api.MijpegError) as e:
        d2.finish_batch_device()
    assert e.value.code == api.ERR_NOT_AVAILABLE and "damaged" in e.value.message
    hd = api.Decoder(None)
    with pytest.raises(api.MijpegError) as e:
        hd.read(bytes(bad))
    _, ref_err, _ = oracle.decode_status(bytes(bad))
    assert e.value.code == ref_err and ref_err < 0  # the verdict of the host walk is the reference's
    hd.close()
    # 5. no restart markers: decoded by the time submit returns, same pixels
    plain = [synth.synth_jpeg(640, 400, 400 + i, 85, "420", 0) for i in range(2)]
    out2 = torch.zeros((2, 400, 640 * 3), dtype=torch.uint8, device="cuda")
    d2.submit_batch_device(plain, 1)
    d2.reconstruct_batch_device(out2.data_ptr(), 400 * 640 * 3, 640 * 3)
    for i in range(2):
        assert np.array_equal(out2[i].cpu().numpy().reshape(400, 640, 3), oracle.decode(plain[i]))
    d.close()
    d2.close()


@pytest.mark.gpu
def test_full_duplex_download_delivers_every_frame(oracle):
    """BatchShard.run(download_to=pinned): each chunk's pixels leave for host memory behind its own reconstruction kernel
    (mijpeg_stream_wait) while later chunks are uploaded and decoded; when run() returns every frame is in host memory and
    equals the oracle's decode.  Two passes over the same buffers (the second overwrites the first's frames in order)."""
    import torch

    cfg = dict(batch.CONFIG4, width=640, height=368, frames=24)
    streams = batch.make_streams(range(24), cfg)
    shard = batch.BatchShard([streams[i] for i in range(24)], 0, chunk=5, depth=3)
    host = torch.zeros(tuple(shard.out.shape), dtype=torch.uint8).pin_memory()
    for _ in range(2):
        host.zero_()
        shard.run(download_to=host)
        for i in range(24):
            exp = oracle.decode(streams[i])
            assert np.array_equal(host[i].numpy().reshape(exp.shape), exp), i
    shard.close()


@pytest.mark.gpu
def test_speculative_reconstruction_is_validated(oracle):
    """MIJPEG_FLAG_SPECULATIVE: the reconstruction of a submitted batch is launched behind its Huffman kernel on the range check
    of the last finished batch of that shape and those tables; mijpeg_finish_batch_device validates.  (1) the assumption holds:
    nothing is redone; (2) a batch whose chroma leaves the assumed gate: the validation reconstructs again -- the pixels are the
    oracle's either way; (3) a damaged member: the validation reports NOT_AVAILABLE like an ordinary finish; (4) the
    pipeline of libjpeg_amd/batch.py on top of it, second pass speculative, with the full-duplex download."""
    import torch
    from PIL import Image
    import io

    from libjpeg_amd import api, synth

    w, h = 416, 240
    row = w * 3

    def encode(img):
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, "JPEG", quality=85, subsampling=2, restart_marker_blocks=2)
        return buf.getvalue()

    rng = np.random.default_rng(4)
    calm = [encode(np.clip(128 + 40 * np.sin(np.arange(w)[None, :, None] / (9.0 + i)) + rng.normal(0, 3, (h, w, 3)), 0, 255).astype(np.uint8)) for i in range(4)]
    wild = [encode(rng.integers(0, 256, (h, w, 3)).astype(np.uint8)) for i in range(4)]  # noise in every channel: chroma far beyond the packed kernel's gate
    out = torch.zeros((4, h, row), dtype=torch.uint8, device="cuda")
    d = api.Decoder(0)

    def pixels(i):
        return out[i].cpu().numpy().reshape(h, w, 3)

    # the hint: one ordinary batch of the calm material
    d.submit_batch_device(calm, 1)
    d.reconstruct_batch_device(out.data_ptr(), h * row, row)
    # (1) holds
    d.submit_batch_device(calm, 1)
    out.zero_()
    d.reconstruct_batch_device(out.data_ptr(), h * row, row, flags=api.FLAG_SPECULATIVE, sync=False)
    d.finish_batch_device()
    again, launched, redone = d.batch_speculation()
    assert launched == 1 and redone == 0 and not again
    for i in range(4):
        assert np.array_equal(pixels(i), oracle.decode(calm[i])), i
    # (2) the wild batch on the calm hint: redone
    d.submit_batch_device(wild, 1)
    out.zero_()
    d.reconstruct_batch_device(out.data_ptr(), h * row, row, flags=api.FLAG_SPECULATIVE, sync=False)
    info = d.finish_batch_device()
    again, launched, redone = d.batch_speculation()
    assert launched == 2 and redone == 1 and again, (launched, redone, list(info.range_max))
    for i in range(4):
        assert np.array_equal(pixels(i), oracle.decode(wild[i])), i
    # ... mijpeg_synchronize validates as well
    d.submit_batch_device(calm, 1)  # (hint is the wild one now: calm lies below it, nothing to redo)
    out.zero_()
    d.reconstruct_batch_device(out.data_ptr(), h * row, row, flags=api.FLAG_SPECULATIVE, sync=False)
    d.synchronize()
    assert d.batch_speculation()[1:] == (3, 1)
    for i in range(4):
        assert np.array_equal(pixels(i), oracle.decode(calm[i])), i
    # (3) a damaged member
    bad = bytearray(calm[1])
    pos = bad.find(b"\xff\xda") + 200
    while any(x == 0xFF for x in bad[pos - 1:pos + 14]):
        pos += 1
    bad[pos:pos + 12] = b"\xff\x00" * 6
    d.submit_batch_device([calm[0], bytes(bad), calm[2]], 1)
    d.reconstruct_batch_device(out.data_ptr(), h * row, row, flags=api.FLAG_SPECULATIVE, sync=False)
    with pytest.raises(api.MijpegError) as e:
        d.finish_batch_device()
    assert e.value.code == api.ERR_NOT_AVAILABLE
    d.close()
    # (4) the pipeline
    cfg = dict(batch.CONFIG4, width=640, height=368, frames=24)
    streams = batch.make_streams(range(24), cfg)
    shard = batch.BatchShard([streams[i] for i in range(24)], 0, chunk=5, depth=3)
    shard.speculate = True  # (opt-in: MIJPEG_BATCH_SPECULATION)
    host = torch.zeros(tuple(shard.out.shape), dtype=torch.uint8).pin_memory()
    for p in range(3):
        host.zero_()
        shard.run(download_to=host)
        for i in range(24):
            exp = oracle.decode(streams[i])
            assert np.array_equal(host[i].numpy().reshape(exp.shape), exp), (p, i)
    assert sum(dec.batch_speculation()[1] for dec in shard.decoders) >= 8 and shard.redone == 0
    shard.close()


@pytest.mark.gpu
def test_cxx_client_of_the_batch_pipeline(oracle, tmp_path):
    """Config 4's driver is behind the C ABI (mijpeg_batch_pipeline_create / _run / _destroy): tests/cxx/batch_client.cpp, built
    against include/mijpeg.h alone, sends 24 streams -- one of them damaged in its entropy coded data, one without restart markers
    -- through the pipeline, pixels into HBM and full duplex into pinned host memory; every frame's hash equals the oracle's
    decode, the chunk with the damaged stream fell back (blocking batch call, then stream by stream on the host), nothing else did."""
    import subprocess

    from libjpeg_amd import synth

    cfg = dict(batch.CONFIG4, width=640, height=368, frames=24)
    streams = batch.make_streams(range(24), cfg)
    blobs = [streams[i] for i in range(24)]
    bad = bytearray(blobs[7])
    pos = bad.find(b"\xff\xd3", bad.find(b"\xff\xda"))
    del bad[pos:pos + 2]  # a restart marker gone: the reference resynchronises at the next one (entropyparser.cpp:117-201) -- the host walk's business
    blobs[7] = bytes(bad)
    assert oracle.decode(blobs[7]).shape == (368, 640, 3)
    blobs[20] = synth.synth_jpeg(640, 368, 77, 85, "420", 0)  # no restart markers: virtual restart intervals from the device walk
    paths = []
    for i, b in enumerate(blobs):
        p = tmp_path / f"{i}.jpg"
        p.write_bytes(b)
        paths.append(str(p))
    exe = str(tmp_path / "batch_client")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    subprocess.run(["g++", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(rocm, "include"), "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cxx", "batch_client.cpp"), "-o", exe, "-L", os.path.join(ROOT, "libjpeg_amd"), "-lmijpeg",
                    "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "libjpeg_amd")], check=True)
    r = subprocess.run([exe, "5", "3", "0", "2"] + paths, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    lines = r.stdout.strip().splitlines()
    assert lines[-1].startswith("summary frames 24 640x368 chunks 5 fallbacks "), lines[-1]
    assert int(lines[-1].split("fallbacks ")[1].split()[0]) == 1  # the chunk of frame 7 (frame 20's takes the device walk)

    def fnv(a):
        h = 1469598103934665603
        for byte in a.tobytes():
            h = ((h ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h

    for i in (0, 6, 7, 8, 19, 20, 23):
        assert int(lines[i].split()[2], 16) == fnv(oracle.decode(blobs[i])), i
