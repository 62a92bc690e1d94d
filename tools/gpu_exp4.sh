cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_batch4k.py tests/test_gpu_parity.py -m gpu -x -q -k "batch or config4" 2>&1 | tail -4
SETTINGS=32x1,32x2,32x3,16x2,16x3,16x4,64x2,8x4 STEPS=3 timeout 200 python tools/batch4k_bench.py 2>&1 | grep -v amdgpu
