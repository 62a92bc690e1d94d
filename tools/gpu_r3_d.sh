#!/bin/bash
# Round 3, visit D: the whole GPU suite, the bench line as the driver runs it, the rocprofv3 passes for profiles/r03
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r3d; export TMPDIR=/tmp
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3d/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/r3d/pytest_gpu.log
echo "== bench"; ( time timeout 900 python bench.py > gpurun_out/r3d/bench.json 2> gpurun_out/r3d/bench.err ); echo "bench exit $?"; tail -c 1500 gpurun_out/r3d/bench.json; tail -5 gpurun_out/r3d/bench.err
echo "== profile"; bash tools/gpu_profile.sh r03 > gpurun_out/r3d/profile.log 2>&1; tail -40 gpurun_out/r3d/profile.log
