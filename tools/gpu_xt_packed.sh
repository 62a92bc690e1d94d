#!/bin/bash
# Round 6, last session: fusedxt420_kernel<true> (legacy chroma filtered as 16-bit pairs inside fused420p_kernel's gate).
# The JPEG XT GPU tests, then an alternating A-B against the 32-bit filters (MIJPEG_NO_XT_PACKED=1). -> gpurun_out/xtpk
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/xtpk; export TMPDIR=/tmp
O=gpurun_out/xtpk
timeout 1500 python -m pytest tests -m gpu -q -x -k "xt or XT or hdr or residual or hidden" > $O/pytest_xt.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_xt.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/ranges.txt
import sys; sys.path.insert(0, ".")
import bench
from libjpeg_amd import api, synth
from oracle import oracle as O
W, H = bench.SIZES["4k"]
data = O.reference_encode_hdr(synth.synth_hdr(W, H, 99), bench.XT_ARGS)
d = api.Decoder(0); f = d.read(data, entropy="host")
print("legacy frame range_max", list(f.range_max)[:3], "kernel", api.kernel_name(f, xt=d.xt_params()))
PY
: > $O/ab.txt
for round in 1 2 3; do
  for off in 0 1; do
    if [ $off = 1 ]; then export MIJPEG_NO_XT_PACKED=1; else unset MIJPEG_NO_XT_PACKED; fi
    echo "round $round MIJPEG_NO_XT_PACKED=${MIJPEG_NO_XT_PACKED:-} $(timeout 300 python tools/xt_launches.py --time --launches 40 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O/ab.txt
  done
done
unset MIJPEG_NO_XT_PACKED
cat $O/ab.txt
