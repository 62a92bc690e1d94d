#!/bin/bash
# A-B of whole-file scheduler strategies (tools/ab/build_variant.sh NAME -mllvm -amdgpu-sched-strategy=...): headline / dense / reference-encoded
# fractions, config 5's two launches, the 12-bit and tile layouts -- default library against tools/ab/libmijpeg_NAME.so.  -> gpurun_out/sched/
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/sched; export TMPDIR=/tmp
O=gpurun_out/sched/ab.txt; : > $O
for round in 1 2; do
  for v in "" "$@"; do
    if [ -n "$v" ]; then export MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_$v.so; else unset MIJPEG_LIBRARY; fi
    echo "== round $round ${v:-default}" >> $O
    python bench.py --no-xt --no-end-to-end --no-cpu-baseline --workload headline 2>/dev/null | grep -o "\"kernel_frac\": [0-9.]*\|\"dense_frac[^}]*\|\"verified\": [a-z]*" | tr "\n" " " >> $O; echo >> $O
    echo "r12: $(timeout 300 python tools/xt_launches.py --time --launches 40 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O
    echo "rR4: $(timeout 300 python tools/xt_launches.py --hidden --time --launches 40 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O
    LAYOUTS=420_12,444_12,422_12,444,422,3x1,411 timeout 600 python tools/layout_bench.py 2>&1 | grep "Gpixel" | cut -c1-200 >> $O
  done
done
unset MIJPEG_LIBRARY
cat $O
