#!/bin/bash
# SQ / LDS / HBM counters of the 12-bit fused kernels (tools/layout_bench.py)
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export LAYOUTS=${LAYOUTS:-420_12,422_12,444_12}
timeout 500 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d /tmp/t1 -o t -- python $R/tools/layout_bench.py > /tmp/t1.log 2>&1
timeout 500 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d /tmp/t2 -o t -- python $R/tools/layout_bench.py > /tmp/t2.log 2>&1
timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/t3 -o t -- python $R/tools/layout_bench.py > /tmp/t3.log 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/t4 -o t -- python $R/tools/layout_bench.py > /tmp/t4.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("/tmp/t1", "/tmp/t2", "/tmp/t3", "/tmp/t4"):
    f = glob.glob(d + "/**/t_counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "fused4" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:60], r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, {n: "%.4g" % (sum(x) / len(x)) for n, x in sorted(v.items())})
PY
grep -v "^$" /tmp/t1.log | tail -12
