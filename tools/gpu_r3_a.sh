#!/bin/bash
# Round 3, visit A: regression of the GPU suite, the bench line's new config 5 section, timeline of the config 4 pipeline.
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r3a; export TMPDIR=/tmp
echo "== pytest gpu"; timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/r3a/pytest_gpu.log
echo "== bench (headline + xt)"; timeout 600 python bench.py --steps 10 --no-traffic --no-dense --no-end-to-end --workload headline > gpurun_out/r3a/bench_xt.json 2> gpurun_out/r3a/bench_xt.err; echo "bench exit $?"; tail -c 3000 gpurun_out/r3a/bench_xt.json
echo "== batch4k plain"; ( cd /tmp; CFG_FRAMES=256 SETTINGS=24x4,16x4,32x3,16x6 STEPS=4 timeout 300 python $ROOT/tools/batch4k_bench.py 2>&1 | grep -v amdgpu.ids | tee $ROOT/gpurun_out/r3a/batch4k_plain.txt )
echo "== batch4k trace"; ( cd /tmp; CFG_FRAMES=256 SETTINGS=24x4 STEPS=2 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $ROOT/gpurun_out/r3a/trace -o t -- python $ROOT/tools/batch4k_bench.py > $ROOT/gpurun_out/r3a/trace.log 2>&1; echo "trace exit $?" )
find gpurun_out/r3a/trace -name "*.csv" | head; du -sh gpurun_out/r3a
