#!/bin/bash
# Alternating A-B of config 5's -r12 launch (tools/xt_launches.py --time) between the default library and tools/ab/libmijpeg_<name>.so
# for every name given; the JPEG XT GPU tests on each variant first.  -> gpurun_out/xtab/
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/xtab; export TMPDIR=/tmp
O=gpurun_out/xtab; : > $O/ab.txt
for v in "$@"; do
  MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "xt or XT or hdr or residual or hidden" > $O/pytest_$v.log 2>&1
  echo "$v: pytest exit $? $(tail -1 $O/pytest_$v.log)" | tee -a $O/ab.txt
done
for round in 1 2 3; do
  for v in "" "$@"; do
    if [ -n "$v" ]; then export MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_$v.so; else unset MIJPEG_LIBRARY; fi
    echo "round $round ${v:-default}: $(timeout 300 python tools/xt_launches.py --time --launches 40 2>&1 | grep -v amdgpu.ids | tail -1)" >> $O/ab.txt
  done
done
unset MIJPEG_LIBRARY
cat $O/ab.txt
