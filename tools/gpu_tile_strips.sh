#!/bin/bash
# A-B of fused_tile_kernel's phase B: one line of an 8-pixel group per lane and iteration (TILE_STRIP 1, the product) against strips
# of 2 and 4 lines per lane (tools/ab/build_variant.sh strip2 -DTILE_STRIP=2, strip4 -DTILE_STRIP=4), alternating, two rounds
L=${LAYOUTS:-3x1,1x4,lumasub,3x3,cmyk}
for r in 1 2; do
  for lib in "" tools/ab/libmijpeg_strip2.so tools/ab/libmijpeg_strip4.so; do
    echo "== round $r ${lib:-product (TILE_STRIP 1)}"
    MIJPEG_LIBRARY=${lib:+$PWD/$lib} LAYOUTS=$L timeout 300 python tools/layout_bench.py 2>&1 | grep "ms/launch" | cut -c1-20,70-125
  done
done
