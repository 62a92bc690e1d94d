#!/bin/bash
# Round 5: the host side of config 5 with hidden bits on the GPU box's cores (no kernel runs)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5y
nproc > gpurun_out/r5y/host_refine_times.txt; lscpu | grep -E "Model name|Socket|Core|Thread|L2|L3|NUMA node\(s\)" >> gpurun_out/r5y/host_refine_times.txt
(echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>&1)"; grep -E "throttled|nr_periods" /sys/fs/cgroup/cpu.stat 2>&1) >> gpurun_out/r5y/host_refine_times.txt
timeout 600 python tools/host_refine_times.py >> gpurun_out/r5y/host_refine_times.txt 2>&1
(echo "after:"; grep -E "throttled|nr_periods" /sys/fs/cgroup/cpu.stat 2>&1) >> gpurun_out/r5y/host_refine_times.txt
tail -60 gpurun_out/r5y/host_refine_times.txt
