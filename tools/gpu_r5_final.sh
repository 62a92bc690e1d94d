#!/bin/bash
# Round 5, measurement visit: profiler passes over the bench command, the bench line, the whole GPU suite
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5final; export TMPDIR=/tmp
O=gpurun_out/r5final
bash tools/gpu_profile.sh r05 > $O/profile.log 2>&1; tail -30 $O/profile.log
cd "$ROOT"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ); echo "bench exit $?"; tail -c 600 $O/bench.json; echo
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log
