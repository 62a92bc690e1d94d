#!/bin/bash
# Quick A/B of fused-kernel variants: MIJPEG_F420_VARIANT=n python bench.py ... for each n given.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for v in "$@"; do
  echo "== variant $v"
  MIJPEG_F420_VARIANT=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('  kernel_ms %.4f  achieved %.1f GB/s  frac %.4f  value %.0f Mpix/s'%(j['roofline']['kernel_ms'],j['roofline']['achieved'],j['roofline']['frac'],j['value']))
"
done
