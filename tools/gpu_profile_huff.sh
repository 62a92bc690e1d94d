#!/bin/bash
# SQ counters of the on-device Huffman kernel (tools/entropy_bench.py, one 8K 4:2:0 frame).  Output: gpurun_out/prof_<tag>/
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-huff}"
OUT="$ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
CMD="python $ROOT/tools/entropy_bench.py"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- $CMD > "$OUT/trace.log" 2>&1; echo "trace exit $?"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/pmc_sq" -o t -- $CMD > "$OUT/pmc_sq.log" 2>&1; echo "pmc_sq exit $?"
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_lds" -o t -- $CMD > "$OUT/pmc_lds.log" 2>&1; echo "pmc_lds exit $?"
cd "$ROOT"
python tools/summarize_prof.py "$OUT" | tee "$OUT/summary.txt"
