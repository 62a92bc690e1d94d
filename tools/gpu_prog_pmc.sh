#!/bin/bash
# huffman_prog_kernel: instruction counters per launch (rocprofv3 --pmc, one pass per counter group), config 5 -rR 4 -z 8
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/progpmc; export TMPDIR=/tmp
O=$ROOT/gpurun_out/progpmc
i=0
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH"; do
  i=$((i+1))
  ( cd /tmp && timeout 200 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $O/g$i -o ms -- python "$ROOT/tools/multiscan_probe.py" run "$ROOT/build/ms" 2 ${STREAMS:-xt4k_rR4_z8} > $O/g$i.log 2>&1 ) || echo "group $i failed: $(tail -2 $O/g$i.log)"
  f=$(find $O/g$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'P'
import csv, sys, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if "huffman_prog" not in r["Kernel_Name"]: continue
    key = (r["Dispatch_Id"], r["Kernel_Name"].split("<")[1].split(">")[0], r["Grid_Size"])
    acc.setdefault(key, {})[r["Counter_Name"]] = acc.setdefault(key, {}).get(r["Counter_Name"], 0) + float(r["Counter_Value"])
for k, v in list(acc.items())[-10:]:
    print(k, {a: int(b) for a, b in v.items()})
P
done
