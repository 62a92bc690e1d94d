#!/bin/bash
# Round 5, measurement visit: profiler passes over the bench command, the bench line, the whole GPU suite, the host side of
# config 5 with hidden bits, a campaign of encoder-written JPEG XT files on the build that is measured
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5final; export TMPDIR=/tmp
O=gpurun_out/r5final
bash tools/gpu_profile.sh r05 > $O/profile.log 2>&1; tail -30 $O/profile.log
cd "$ROOT"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ); echo "bench exit $?"; tail -c 600 $O/bench.json; echo
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log
timeout 300 python tools/host_refine_times.py > $O/host_refine_times.txt 2>&1; tail -12 $O/host_refine_times.txt
N=1200 SEED=20261027 timeout 900 python tools/xt_gpu_campaign.py 2>&1 | grep -v "amdgpu.ids" > $O/xt_gpu_campaign.txt; tail -5 $O/xt_gpu_campaign.txt
