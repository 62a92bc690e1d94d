cd $GRAFT_REPO_ROOT
for g in 4 8 16 32 64; do for l in 64 32 16; do echo "== group $g lanes $l"; MIJPEG_BATCH_GROUP=$g MIJPEG_HUFF_LANES=$l SETTINGS=32x2,64x2,128x1 STEPS=3 timeout 200 python tools/batch4k_bench.py 2>&1 | grep chunk; done; done
