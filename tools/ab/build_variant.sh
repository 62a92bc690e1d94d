#!/bin/bash
# tools/ab/build_variant.sh NAME [-DMACRO=VALUE ...]: kernels.hip compiled with the given macros, linked with the other objects of
# the current build into tools/ab/libmijpeg_NAME.so (git-ignored; travels to the GPU box).  MIJPEG_LIBRARY selects it at run time.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
NAME="$1"; shift
cd "$ROOT/libjpeg_amd/csrc"
make -s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c kernels.hip -o "build/kernels_$NAME.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/ab/libmijpeg_$NAME.so" "build/kernels_$NAME.o" build/forward.o build/hencode.o build/huffman.o \
  build/capi.o build/batch_pipeline.o build/host_decoder.o build/encoder.o build/jpeg_class.o -pthread
rm -f build/kernels_$NAME.o.*
echo "built tools/ab/libmijpeg_$NAME.so"
