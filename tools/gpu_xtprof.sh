#!/bin/bash
# rocprofv3 over config 5's two fused kernels (kernel trace + stats, SQ counters, HBM bytes in separate passes)
TAG="${1:-r06}"
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/prof_xt_${TAG:-r06}"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for v in ${VARIANTS:-"" "--hidden"}; do
  tag=r12; [ -n "$v" ] && tag=rR4
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$tag" -o t -- python $ROOT/tools/xt_launches.py $v > "$OUT/trace_$tag.log" 2>&1; echo "trace $tag exit $?"
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_sq_$tag" -o t -- python $ROOT/tools/xt_launches.py $v > "$OUT/pmc_sq_$tag.log" 2>&1; echo "pmc_sq $tag exit $?"
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_$tag" -o t -- python $ROOT/tools/xt_launches.py $v > "$OUT/pmc_fetch_$tag.log" 2>&1; echo "fetch $tag exit $?"
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_$tag" -o t -- python $ROOT/tools/xt_launches.py $v > "$OUT/pmc_write_$tag.log" 2>&1; echo "write $tag exit $?"
done
cd "$ROOT"
TAG=$TAG python - <<'PY'
import csv, glob, os, collections
import os as _os
out = "gpurun_out/prof_xt_" + _os.environ.get("TAG", "r06")
with open(os.path.join(out, "summary.txt"), "w") as fo:
    for tag in ("r12", "rR4"):
        for f in glob.glob(f"{out}/trace_{tag}/**/*kernel_stats.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "fusedxt" in r["Name"]:
                    print(tag, "kernel stats:", r["Name"][:60], "calls", r["Calls"], "avg ns", r["AverageNs"], "min ns", r["MinNs"], file=fo)
        for kind in ("pmc_sq", "pmc_fetch", "pmc_write"):
            acc = collections.defaultdict(list)
            for f in glob.glob(f"{out}/{kind}_{tag}/**/*counter_collection.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    if "fusedxt" in r.get("Kernel_Name", ""):
                        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in sorted(acc.items()):
                print(tag, kind, k, "n", len(v), "mean %.6g" % (sum(v) / len(v)), file=fo)
print(open(os.path.join(out, "summary.txt")).read())
PY
