#!/bin/bash
# lane utilisation of huffman_scan_kernel (one decoder object, 32 x 4K frames per launch): thread-cycles against wave-instructions
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export CFG_FRAMES=64 SETTINGS=32x1 STEPS=3
rm -rf /tmp/hu1 /tmp/hu2
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d /tmp/hu1 -o t -- python $R/tools/batch4k_bench.py > /tmp/hu1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH --output-format csv -d /tmp/hu2 -o t -- python $R/tools/batch4k_bench.py > /tmp/hu2.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("/tmp/hu1", "/tmp/hu2"):
    fs = glob.glob(d + "/**/t_counter_collection.csv", recursive=True)
    if not fs: print("no counters in", d); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "huffman_scan" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print({k: "%.4g" % (sum(x) / len(x)) for k, x in sorted(acc.items())})
PY
tail -3 /tmp/hu1.log
