#!/bin/bash
# rocprofv3 passes over bench.py: kernel trace + stats, SQ counters, HBM read bytes, HBM write bytes
# (separate --pmc passes as MI355X_MICROARCH.md prescribes).  Output: gpurun_out/prof_<tag>/
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r01}"
OUT="$ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
# the trace pass runs the benchmark as the driver does (64 frames per launch, clock settled first): its average kernel
# duration is what bench.py's roofline.kernel_ms must agree with; the counter passes use short 8-frame launches
TRACE="python $ROOT/bench.py --no-cpu-baseline --no-end-to-end --no-traffic --no-dense --no-xt --workload headline ${BENCH_ARGS:-}"
BENCH="python $ROOT/bench.py --steps 10 --warmup 2 --frames 8 --settle-ms 0 --no-cpu-baseline --no-end-to-end --no-traffic --no-dense --no-xt --workload headline ${BENCH_ARGS:-}"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- $TRACE > "$OUT/trace.log" 2>&1; echo "trace exit $?"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_sq" -o t -- $BENCH > "$OUT/pmc_sq.log" 2>&1; echo "pmc_sq exit $?"
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_lds" -o t -- $BENCH > "$OUT/pmc_lds.log" 2>&1; echo "pmc_lds exit $?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o t -- $BENCH > "$OUT/pmc_fetch.log" 2>&1; echo "fetch exit $?"
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o t -- $BENCH > "$OUT/pmc_write.log" 2>&1; echo "write exit $?"
cd "$ROOT"
python tools/summarize_prof.py "$OUT" | tee "$OUT/summary.txt"
