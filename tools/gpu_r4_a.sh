#!/bin/bash
# Round 4, visit a: the whole GPU suite on the new kernels, the headline launch A-B against the previous build
# (tools/ab/libmijpeg_base.so, built from the previous commit's kernels.hip), the host decode of config 5's -rR 4 variant with
# its switches, and the bench line.
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r4a; export TMPDIR=/tmp
O=gpurun_out/r4a
echo "== pytest gpu"; timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 $O/pytest_gpu.log
HL="--workload headline --no-cpu-baseline --no-end-to-end --no-traffic --no-xt --steps 20 --warmup 3"
for rep in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_base.so; else unset MIJPEG_LIBRARY; fi
    timeout 300 python bench.py $HL > $O/hl_${v}_$rep.json 2> $O/hl_${v}_$rep.err
    python - "$O/hl_${v}_$rep.json" $v $rep <<'PY'
import json, sys
try:
    r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[2], sys.argv[3], "headline", r["roofline"]["frac"], r["roofline"]["kernel_ms"], "dense", r["roofline_dense"]["dense"]["frac"], "beyond", r["roofline_dense"]["beyond_gate"]["frac"],
          "refenc", r.get("roofline_reference_encoded", {}).get("frac"))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "failed", e)
PY
  done
done
unset MIJPEG_LIBRARY
echo "== xt host decode"; timeout 600 python tools/xt_host_bench.py 1 16 64 > $O/xt_host.txt 2>&1; cat $O/xt_host.txt
echo "== bench"; ( time timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err ); echo "bench exit $?"; tail -c 300 $O/bench.json; tail -3 $O/bench.err
