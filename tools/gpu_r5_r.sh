#!/bin/bash
# Round 5: the 16-bit (v_dot2) second pass of fused420p_kernel -- parity at its gate, then headline / dense / reference-encoded with and without
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5r; export TMPDIR=/tmp
O=gpurun_out/r5r
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "extreme or pruned or vs_oracle or adversarial or random or headline or 420" > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest_gpu.log
B="python bench.py --no-cpu-baseline --no-end-to-end --no-traffic --no-xt --workload headline"
for round in 1 2 3; do
  for v in dot2 nodot2; do
    if [ $v = nodot2 ]; then export MIJPEG_NO_DOT2=1; else unset MIJPEG_NO_DOT2; fi
    timeout 600 $B 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v round $round: headline %.4f (%.3f ms) dense %.4f beyond_gate %.4f reference-encoded %.4f verified %s' % (j['roofline']['frac'], j['roofline']['kernel_ms'], j['roofline_dense']['dense']['frac'], j['roofline_dense']['beyond_gate']['frac'], j['roofline_reference_encoded']['frac'], j.get('verified')))" | tee -a $O/dot2_ab.txt
  done
done
