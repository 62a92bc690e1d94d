import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from libjpeg_amd import api, synth, pipeline
W, H = 7680, 4320
jp = [synth.synth_jpeg(W, H, 1234 + i, 85, "420", 8) for i in range(2)]
for depth in (1, 2, 3, 4):
    p = pipeline.FramePipeline(0, depth=depth)
    p.run([jp[i % 2] for i in range(2 * depth)])
    t = time.perf_counter(); n = 16; p.run([jp[i % 2] for i in range(n)]); dt = time.perf_counter() - t
    tm = [d.timing() for d in p._decoders]
    print("depth", depth, "ms/frame %.2f" % (dt / n * 1e3), {k: round(v * 1e3, 2) for k, v in tm[0].items()})
    p.close()
