#!/bin/bash
# fused420p_kernel with LDS-staged, fully coalesced pixel stores (tools/ab/build_variant.sh staged -DF420P_STAGED=1): the GPU parity file on the
# variant, then bench.py's headline / dense / reference-encoded fractions alternating with the default library.  -> gpurun_out/staged/
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
V="${1:-staged}"
cd "$ROOT"; mkdir -p gpurun_out/staged; export TMPDIR=/tmp
O=gpurun_out/staged/ab.txt; : > $O
timeout 200 tools/microbench/stream_ceiling --staged >> $O 2>&1
MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_$V.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_batch.py -m gpu -q -x > gpurun_out/staged/pytest_$V.log 2>&1
echo "$V: pytest exit $? $(tail -n 1 gpurun_out/staged/pytest_$V.log)" | tee -a $O
for round in 1 2 3; do
  for v in "" "$V"; do
    if [ -n "$v" ]; then export MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_$v.so; else unset MIJPEG_LIBRARY; fi
    echo -n "round $round ${v:-default}: " >> $O
    python bench.py --no-xt --no-end-to-end --no-cpu-baseline --workload headline 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*\|\"kernel_frac\": [0-9.]*\|\"dense_frac[^}]*\|\"verified\": [a-z]*" | tr "\n" " " >> $O; echo >> $O
  done
done
unset MIJPEG_LIBRARY
cat $O
