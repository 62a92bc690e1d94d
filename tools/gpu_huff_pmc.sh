cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/tools/batch4k_bench.py"
export CFG_FRAMES=64 SETTINGS=32x1 STEPS=2
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d /tmp/h1 -o t -- $CMD > /tmp/h1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/h2 -o t -- $CMD > /tmp/h2.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("/tmp/h1", "/tmp/h2"):
    f = glob.glob(d + "/**/t_counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "huffman_scan" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print("%-24s n=%d mean=%.4g" % (k, len(v), sum(v) / len(v)))
PY
