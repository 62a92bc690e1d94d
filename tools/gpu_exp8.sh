cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_batch4k.py tests/test_gpu_parity.py -m gpu -q -x -k "submit or without_restart or walk or batch" 2>&1 | tail -3
CFG_8K=1 CFG_DRI=0 CFG_FRAMES=16 SETTINGS=16x1,8x2,4x3,4x4,2x4 STEPS=3 timeout 300 python tools/batch4k_bench.py 2>&1 | grep -v amdgpu
CFG_DRI=0 CFG_FRAMES=128 SETTINGS=32x2,16x3,16x4,64x2 STEPS=3 timeout 300 python tools/batch4k_bench.py 2>&1 | grep -v amdgpu
