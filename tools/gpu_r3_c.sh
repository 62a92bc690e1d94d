#!/bin/bash
# Round 3, visit C: fused_tile_kernel (tests + layouts), headline LDS-store experiment, config 4 after the Huffman tables
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"; mkdir -p gpurun_out/r3c; export TMPDIR=/tmp
echo "== pytest gpu (fused tile first)"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_tile" > gpurun_out/r3c/pytest_tile.log 2>&1; echo "exit $?"; tail -15 gpurun_out/r3c/pytest_tile.log
echo "== pytest gpu (all)"; timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3c/pytest_gpu.log 2>&1; echo "exit $?"; tail -5 gpurun_out/r3c/pytest_gpu.log
cd /tmp
echo "== layouts, one pass"; LAYOUTS=cmyk,3x1,1x4,lumasub,3x3,444_12,422_12 timeout 300 python $ROOT/tools/layout_bench.py 2>&1 | grep -v amdgpu.ids | tee $ROOT/gpurun_out/r3c/layouts_tile.txt
echo "== layouts, generic pair"; MIJPEG_NO_FUSED_TILE=1 LAYOUTS=cmyk,3x1,1x4,lumasub,3x3,444_12,422_12 timeout 300 python $ROOT/tools/layout_bench.py 2>&1 | grep -v amdgpu.ids | tee $ROOT/gpurun_out/r3c/layouts_pair.txt
echo "== headline: as built / without the 16-bit chroma stores of phase A (experiment, wrong pixels)"
for lib in "" "$ROOT/tools/exp/libmijpeg_nostores.so"; do
  MIJPEG_LIBRARY=$lib timeout 300 python $ROOT/bench.py --steps 20 --no-traffic --no-dense --no-end-to-end --no-xt --no-cpu-baseline --workload headline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', r['roofline']['kernel_ms'], r['roofline']['frac'])"
done | tee $ROOT/gpurun_out/r3c/headline_exp.txt
echo "== batch4k plain"; CFG_FRAMES=256 SETTINGS=32x4r,24x4r,40x4r,32x5r STEPS=4 timeout 300 python $ROOT/tools/batch4k_bench.py 2>&1 | grep -v amdgpu.ids | tee $ROOT/gpurun_out/r3c/batch4k_plain.txt
