import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from libjpeg_amd import api, synth
for (w, h, sub, dri) in ((64, 48, "444", 1), (200, 120, "420", 1), (200, 120, "420", 8), (640, 368, "420", 4), (3840, 2160, "420", 8)):
    data = synth.synth_jpeg(w, h, 5, 85, sub, dri)
    a, b = api.Decoder(0), api.Decoder(0)
    fa = a.read(data, entropy="host")
    try:
        fb = b.read(data, entropy="gpu")
    except api.MijpegError as e:
        print(w, h, sub, dri, "gpu refused", e); continue
    for c in range(fa.components):
        ca, cb = a.coefficients(c), b.coefficients(c)
        if not np.array_equal(ca, cb):
            bad = np.argwhere((ca != cb).any(axis=2))
            print(w, h, sub, dri, "comp", c, "blocks differing", len(bad), "of", ca.shape[0]*ca.shape[1], "first", bad[:5].tolist())
            by, bx = bad[0]
            print("  host", ca[by, bx][:16].tolist()); print("  gpu ", cb[by, bx][:16].tolist())
            break
    else:
        print(w, h, sub, dri, "ok")
