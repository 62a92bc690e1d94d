#!/bin/bash
# Round 4, visit l: the profiler passes over the headline launch (tools/gpu_profile.sh r04) and the driver-shaped bench line
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r4l; export TMPDIR=/tmp
O=gpurun_out/r4l
bash tools/gpu_profile.sh r04 > $O/profile.log 2>&1; tail -5 $O/profile.log
timeout 300 python bench.py --no-cpu-baseline --no-end-to-end --no-xt --workload headline > $O/bench_profile_visit.json 2> $O/bench_profile_visit.err; tail -c 400 $O/bench_profile_visit.json
echo "== bench"; ( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ); echo "bench exit $?"; tail -c 600 $O/bench.json; tail -3 $O/bench.err
