"""Frame pipeline for the batch case (BASELINE config 4): several decoder objects per GPU, each with its own HIP
stream, fed by a small thread pool, so that the host entropy decoding of frame i+1, the H2D/D2H copies and the kernel
of frame i overlap.  Frames are independent: nothing is exchanged, ranks take frames r, r+N, ... (sharding.py).

Measured on MI355X / PCIe Gen5 (tools/pipe_bench.cpp, 8K 4:2:0): 5.9 ms per frame with one decoder, 4.4-4.9 ms with two,
no gain beyond: a frame moves ~100 MB of int16 coefficients up and ~100 MB of pixels down, and the two directions do
not overlap in practice, so ~4 ms of PCIe time per 8K frame is the floor of the host-entropy-decode design.  Streams
with restart markers are entropy-decoded on the device instead (entropy="auto"): 5.6 MB go up, the pixels come down.

The ctypes calls release the GIL; the host Huffman workers are a process-wide pool inside libmijpeg.so that serves one
frame at a time, the copies and kernels of the other in-flight frames proceed on their streams meanwhile."""
from __future__ import annotations

import queue
import threading
from typing import Callable, Iterable, Iterator

import numpy as np

from . import api


class FramePipeline:
    def __init__(self, device: int = 0, depth: int = 2, threads: int = 0, entropy: str = "auto"):
        """entropy: "host", "gpu" or "auto" (Decoder.read): with restart markers in the stream the Huffman decoding
        runs on the device and only the compressed bytes cross PCIe upwards."""
        self.device = device
        self.depth = max(1, depth)
        self.threads = threads
        self.entropy = entropy
        self._decoders = [api.Decoder(device) for _ in range(self.depth)]

    def close(self):
        for d in self._decoders:
            d.close()
        self._decoders = []

    def run(self, streams: Iterable[bytes], sink: Callable[[int, np.ndarray], None] | None = None,
            reuse_buffers: bool = True) -> int:
        """Decode every stream; sink(index, pixels) is called from worker threads as frames complete (pixels are only
        valid during the call when reuse_buffers is set).  Returns the number of frames."""
        work: "queue.Queue[tuple[int, bytes] | None]" = queue.Queue(maxsize=2 * self.depth)
        errors: list[BaseException] = []
        count = [0]
        lock = threading.Lock()

        def worker(dec: api.Decoder):
            buf = None
            try:
                while True:
                    item = work.get()
                    if item is None:
                        return
                    idx, data = item
                    info = dec.read(data, self.threads, self.entropy)
                    shape = (info.height, info.width, info.components)
                    dtype = np.uint8 if info.sample_bytes <= 1 else np.uint16
                    if not reuse_buffers or buf is None or buf.array.shape != shape or buf.array.dtype != dtype:
                        if buf is not None:
                            buf.close()
                        buf = api.PinnedFrame(shape[0], shape[1], shape[2], dtype)  # D2H lands here directly
                    dec.reconstruct_into(buf.array)
                    if sink is not None:
                        sink(idx, buf.array)
                    with lock:
                        count[0] += 1
            except BaseException as e:  # noqa: BLE001 - reported to the caller below
                errors.append(e)
                # keep draining so the producer does not block forever
                while work.get() is not None:
                    pass
            finally:
                if buf is not None:
                    buf.close()

        ts = [threading.Thread(target=worker, args=(d,), daemon=True) for d in self._decoders]
        for t in ts:
            t.start()
        for i, data in enumerate(streams):
            if errors:
                break
            work.put((i, data))
        for _ in ts:
            work.put(None)
        for t in ts:
            t.join()
        if errors:
            raise errors[0]
        return count[0]


def decode_batch(streams: Iterable[bytes], device: int = 0, depth: int = 2) -> Iterator[np.ndarray]:
    """Convenience: decoded frames in input order (copies)."""
    results: dict[int, np.ndarray] = {}
    p = FramePipeline(device, depth)
    try:
        n = p.run(streams, lambda i, px: results.__setitem__(i, px.copy()))
    finally:
        p.close()
    for i in range(n):
        yield results[i]
