cd $GRAFT_REPO_ROOT
for g in 32 64; do echo "== group $g"; MIJPEG_BATCH_GROUP=$g SETTINGS=32x1,32x2,32x3,64x1,64x2,16x4 STEPS=3 timeout 200 python tools/batch4k_bench.py 2>&1 | grep -v amdgpu; done
