#!/bin/bash
# fused422_kernel with staged stores (default) against tools/ab/libmijpeg_nostg422.so (-DF422_STAGED=0): the parity file on the default build,
# then tools/layout_bench.py alternating (plain and SATURATED=1 = the wide flavour).  -> gpurun_out/stg422/
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/stg422; export TMPDIR=/tmp
O=gpurun_out/stg422/ab.txt; : > $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/stg422/pytest.log 2>&1; echo "pytest exit $? $(tail -n 1 gpurun_out/stg422/pytest.log)" | tee -a $O
for round in 1 2 3; do
  for v in "" nostg422; do
    if [ -n "$v" ]; then export MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_$v.so; else unset MIJPEG_LIBRARY; fi
    echo "== round $round ${v:-default (staged)}" >> $O
    LAYOUTS=422 timeout 300 python tools/layout_bench.py 2>&1 | grep Gpixel | cut -c1-200 >> $O
    SATURATED=1 LAYOUTS=422 timeout 300 python tools/layout_bench.py 2>&1 | grep Gpixel | cut -c1-200 >> $O
  done
done
unset MIJPEG_LIBRARY
cat $O
