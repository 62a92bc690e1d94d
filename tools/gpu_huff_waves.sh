cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in ${WAVES:-2 3 4 5 6}; do
  rm -rf /tmp/pw$w
  MIJPEG_HUFF_WAVES=$w CFG_FRAMES=64 SETTINGS=32x1 STEPS=2 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw$w -o t -- python $R/tools/batch4k_bench.py > /tmp/pw$w.log 2>&1
  echo "waves $w: $(find /tmp/pw$w -name '*kernel_stats.csv' -exec grep huffman_scan {} \; | cut -d, -f2-6)"
done
