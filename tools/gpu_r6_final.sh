#!/bin/bash
# Round 6, measurement visit: profiler passes over the bench command, the bench line, the whole GPU suite, config 5's kernels, the
# device decoder of progressive frames / hidden refinement scans
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r6final /tmp/ms; export TMPDIR=/tmp
O=gpurun_out/r6final
bash tools/gpu_profile.sh r06 > $O/profile.log 2>&1; tail -30 $O/profile.log
cd "$ROOT"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err ); echo "bench exit $?"; tail -c 900 $O/bench.json; echo
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 $O/pytest_gpu.log
timeout 200 python tools/multiscan_probe.py make /tmp/ms > /dev/null 2>&1
MIJPEG_TRACE_SUBMIT=1 MIJPEG_READ_TIMES=1 timeout 200 python tools/multiscan_probe.py run /tmp/ms 6 > $O/multiscan_probe.txt 2>&1; grep "ran on" $O/multiscan_probe.txt
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$O/multiscan_prof" -o ms -- python "$ROOT/tools/multiscan_probe.py" run /tmp/ms 4 > "$ROOT/$O/multiscan_prof.log" 2>&1 ); echo "multiscan rocprof exit $?"
f=$(find $O/multiscan_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f"
for i in 1 2; do timeout 120 python tools/xt_launches.py --hidden --time --launches 40 2>&1 | tail -1; MIJPEG_XTW_ONE_WAVE=1 timeout 120 python tools/xt_launches.py --hidden --time --launches 40 2>&1 | tail -1; done | tee $O/xtw_ab.txt
