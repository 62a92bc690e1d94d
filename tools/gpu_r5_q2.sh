#!/bin/bash
# Round 5: the container's CPU quota (cgroup cpu.max) against the host side -- config 5 with hidden bits with back-off waits and
# with yield loops, config 4 at 64 and at 16 pool threads, the throttling counters around each
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r5q2
O=gpurun_out/r5q2/cpu_quota.txt
thr() { echo "$1: $(grep -E 'nr_periods|nr_throttled|throttled_usec' /sys/fs/cgroup/cpu.stat | tr '\n' ' ')"; }
{ echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max)   (quota and period in microseconds); logical CPUs: $(nproc)"; thr start; } > $O
timeout 300 python tools/host_refine_times.py --rounds 1 --reads 16 2>&1 | grep -E "^masks|^blocks " >> $O; thr "after config 5 reads, back-off waits" >> $O
MIJPEG_SPIN_WAITS=1 timeout 300 python tools/host_refine_times.py --rounds 1 --reads 16 2>&1 | grep -E "^masks|^blocks " >> $O; thr "after config 5 reads, yield loops (MIJPEG_SPIN_WAITS=1)" >> $O
for t in 64 16 32; do
  MIJPEG_THREADS=$t timeout 600 python bench.py --workload batch4k --no-xt --no-cpu-baseline --no-end-to-end --no-traffic --no-dense --emulate-world 0 > gpurun_out/r5q2/batch_$t.json 2> gpurun_out/r5q2/batch_$t.err
  python - $t >> $O <<'PY'
import json, sys
t = sys.argv[1]
try:
    b = json.load(open(f"gpurun_out/r5q2/batch_{t}.json"))
    k = b.get("batch4k", b)
    print(f"config 4, MIJPEG_THREADS={t}: ms_per_batch", k.get("ms_per_batch"), "median", k.get("median_ms_per_batch"), "steps", k.get("step_ms"), "chunk", k.get("chunk_frames"), "objects", k.get("decoder_objects"))
except Exception as e:
    print("config 4", t, "failed", repr(e))
PY
  thr "after config 4 at $t threads" >> $O
done
cat $O
