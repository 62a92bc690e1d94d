cd $GRAFT_REPO_ROOT
for b in 0 4 16 0 4 1 32; do MIJPEG_RECT_BAND_MIB=$b timeout 120 python tools/rect_bench.py 2>&1 | grep -v amdgpu; done
