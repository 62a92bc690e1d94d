#!/bin/bash
# Headline launch A-B over the libraries tools/ab/build_variant.sh made: for every tools/ab/libmijpeg_*.so named on the command
# line (default: all) a parity subset of the GPU suite (the packed 4:2:0 kernel's tests), then REPS rounds of the headline bench
# with the variants interleaved.   usage: tools/gpu_hl_variants.sh OUTDIR [name ...]
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; export TMPDIR=/tmp
O=gpurun_out/${1:-variants}; shift; mkdir -p $O
NAMES="$@"; [ -z "$NAMES" ] && NAMES=$(ls tools/ab/libmijpeg_*.so | sed 's/.*libmijpeg_\(.*\)\.so/\1/')
REPS=${REPS:-3}
K="fused420_tile_edges or golden_reference or fused420_packed or full_size or unaligned_output or extreme_coefficients_at_the_packed or pruned_idct or round_trip_properties"
for v in $NAMES; do
  MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$K" > $O/parity_$v.log 2>&1
  echo "parity $v: $(tail -1 $O/parity_$v.log)"
done
HL="--workload headline --no-cpu-baseline --no-end-to-end --no-traffic --no-xt --steps 20 --warmup 3"
for rep in $(seq 1 $REPS); do
  for v in $NAMES; do
    MIJPEG_LIBRARY=$ROOT/tools/ab/libmijpeg_$v.so timeout 300 python bench.py $HL > $O/hl_${v}_$rep.json 2> $O/hl_${v}_$rep.err
    python - "$O/hl_${v}_$rep.json" $v $rep <<'PY'
import json, sys
try:
    r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[2], sys.argv[3], "headline", r["roofline"]["frac"], r["roofline"]["kernel_ms"], "dense", r["roofline_dense"]["dense"]["frac"], "beyond", r["roofline_dense"]["beyond_gate"]["frac"],
          "refenc", r.get("roofline_reference_encoded", {}).get("frac"))
except Exception as e:
    print(sys.argv[2], sys.argv[3], "failed", e)
PY
  done
done | tee $O/headline_ab.txt
