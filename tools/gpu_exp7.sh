cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cat > /tmp/nodri.py <<'PY'
import sys, os, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from libjpeg_amd import api, synth
W, H = 7680, 4320
frames = [synth.encode_jpeg(synth.synth_image(W, H, 2000 + i), 85, "420", restart_mcus=0) for i in range(4)]
n = 16
batch = [frames[i % 4] for i in range(n)]
d = api.Decoder(0)
out = torch.empty((n, H, W * 3), dtype=torch.uint8, device="cuda")
for it in range(3):
    t0 = time.perf_counter()
    d.decode_batch_device(batch)
    d.reconstruct_batch_device(out.data_ptr(), H * W * 3, W * 3)
    print("batch16 no-DRI ms", (time.perf_counter() - t0) * 1e3, d.timing(), d.device_walk_rounds(), flush=True)
PY
cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/pn -o t -- python /tmp/nodri.py 2>&1 | grep "batch16"
cat /tmp/pn/t_kernel_stats.csv | cut -c1-160; cat /tmp/pn/t_memory_copy_stats.csv | cut -c1-120
