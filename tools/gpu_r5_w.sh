#!/bin/bash
# Round 5: 48-byte line pieces with and without the non-temporal hint in the 12-bit 4:2:0 kernel and the two config-5 kernels
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5w; export TMPDIR=/tmp
O=gpurun_out/r5w
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "12bit or 12_bit" > $O/pytest.log 2>&1; echo "pytest exit $?"; tail -2 $O/pytest.log
for round in 1 2; do
for v in "" t420 txt; do
  echo "== variant: ${v:-product}" >> $O/nt_stores.txt
  if [ -n "$v" ]; then export MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_$v.so; else unset MIJPEG_LIBRARY; fi
  LAYOUTS=420_12,444_12,422_12 timeout 600 python tools/layout_bench.py 2>&1 | grep "_12:" | cut -c1-170 >> $O/nt_stores.txt
  timeout 600 python tools/xt_bench.py 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-400 >> $O/nt_stores.txt
done
done
cat $O/nt_stores.txt
