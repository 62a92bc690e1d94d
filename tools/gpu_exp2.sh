cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for cfg in "32 64" "32 16" "4 64"; do set -- $cfg; echo "== group $1 lanes $2";
 (cd /tmp && MIJPEG_BATCH_GROUP=$1 MIJPEG_HUFF_LANES=$2 SETTINGS=32x1 STEPS=2 timeout 200 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/p_$1_$2 -o t -- python $GRAFT_REPO_ROOT/tools/batch4k_bench.py 2>&1 | grep "chunk ")
 head -4 /tmp/p_$1_$2/t_kernel_stats.csv | cut -c1-150; head -3 /tmp/p_$1_$2/t_memory_copy_stats.csv | cut -c1-120
done
