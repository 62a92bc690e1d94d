#!/bin/bash
# Round 6, closing verification: smoke, the bench line, the whole GPU suite (-> gpurun_out/r6verify)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r6verify; export TMPDIR=/tmp
O=gpurun_out/r6verify
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ); echo "bench exit $?"; tail -c 1100 $O/bench.json; echo
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log
