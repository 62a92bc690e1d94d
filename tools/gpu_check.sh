#!/bin/bash
# One GPU-box visit: smoke, parity tests, bench, kernel-trace profile.  Everything lands in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo"; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
echo "== pytest gpu"
timeout ${PYTEST_TIMEOUT:-900} python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/pytest_gpu.log
echo "== bench"
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.log 2>&1; echo "bench exit $?"; tail -3 gpurun_out/bench.log
echo "== rocprof kernel trace"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o trace -- python "$OLDPWD/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end > "$OLDPWD/gpurun_out/prof.log" 2>&1 ); echo "rocprof exit $?"
find gpurun_out/prof -name "*kernel_stats*" | head; for f in $(find gpurun_out/prof -name "*kernel_stats*.csv" | head -1); do head -8 "$f"; done
