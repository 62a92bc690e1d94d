#!/bin/bash
# rocprofv3 kernel statistics of the kernels bench.py does not time: the other fused layouts, JPEG XT, entropy decoding.
# Output: gpurun_out/prof_extra/summary.txt (copy into profiles/)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$ROOT/gpurun_out/prof_extra"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
run() { # tag, command...
  local tag=$1; shift
  rm -rf "$OUT/$tag"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$tag" -o t -- "$@" > "$OUT/$tag.log" 2>&1
  echo "== $tag: $* (exit $?)"
  grep -v amdgpu.ids "$OUT/$tag.log" | grep -E "Gpixel|Mpix|ms per frame|to coefficients|read of one" | head -20
  python - <<PY
import csv, glob
f = glob.glob("$OUT/$tag/**/t_kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    print("   kernel, calls, average us, total ms")
    for r in rows:
        if "rocclr" in r["Name"]: continue
        print("   %-60s %6s %10.1f %10.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
}
run layouts env LAYOUTS=420,422,444,gray python $ROOT/tools/layout_bench.py
run xt python $ROOT/tools/xt_bench.py
run entropy env RI=8 python $ROOT/tools/entropy_bench.py
