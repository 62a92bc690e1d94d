"""Per-launch duration of the bench kernel over a long run (clock ramp / variance study)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from libjpeg_amd import api, synth
W, H, F = 7680, 4320, 8
dec = api.Decoder(0)
data = synth.synth_jpeg(W, H, 1234, 85, "420", 8)
info = dec.read(data)
planes = np.concatenate([dec.coefficients(c).reshape(-1) for c in range(3)])
n = int(info.coef_count)
coef = torch.empty((F, n), dtype=torch.int16, device="cuda")
for f in range(F): coef[f].copy_(torch.from_numpy(planes))
out = torch.empty((F, H, W * 3), dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream()
N = int(os.environ.get("N", "120"))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
ev[0].record(s)
for i in range(N):
    api.launch_reconstruct(info, coef.data_ptr(), out.data_ptr(), F, W * 3, H * W * 3, n, stream=s.cuda_stream)
    ev[i + 1].record(s)
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
print(" ".join(f"{x:.3f}" for x in t))
