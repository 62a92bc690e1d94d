#!/bin/bash
# LDS occupancy experiments on huffman_scan_kernel (round 2): table sharing on / off, another stream ring size.  The ring
# size is a compile-time constant: build a second library first, e.g.
#   (cd libjpeg_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMIJPEG_HUFF_RING=128 -c huffman.hip -o build/huffman_alt.o &&
#    hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmijpeg_r64.so build/kernels.o build/forward.o build/hencode.o build/huffman_alt.o \
#          build/capi.o build/host_decoder.o build/encoder.o build/jpeg_class.o -pthread)
# and the script compares it with the default build through MIJPEG_LIBRARY.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
run() { echo "== $*"; (cd /tmp && env "$@" SETTINGS=32x1 STEPS=2 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$RANDOM -o t -- python $GRAFT_REPO_ROOT/tools/batch4k_bench.py > /tmp/log.txt 2>&1; grep "chunk " /tmp/log.txt; f=$(ls -t /tmp/p_*/t_kernel_stats.csv | head -1); grep huffman_scan $f | cut -d, -f1-4); }
run MIJPEG_HUFF_NO_TABLE_SHARING=1
run A=1
run MIJPEG_LIBRARY=$GRAFT_REPO_ROOT/libjpeg_amd/libmijpeg_r64.so
run MIJPEG_LIBRARY=$GRAFT_REPO_ROOT/libjpeg_amd/libmijpeg_r64.so MIJPEG_HUFF_LANES=32
run MIJPEG_HUFF_LANES=32
echo "== tests with r64"; MIJPEG_LIBRARY=$GRAFT_REPO_ROOT/libjpeg_amd/libmijpeg_r64.so timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "entropy or batch or walk or randomised" 2>&1 | tail -3
echo "== 8K single frame kernel (entropy_bench)"; for lib in libmijpeg.so libmijpeg_r64.so; do MIJPEG_LIBRARY=$GRAFT_REPO_ROOT/libjpeg_amd/$lib RI=8 timeout 200 python tools/entropy_bench.py 2>&1 | grep -v amdgpu | tail -4; done
